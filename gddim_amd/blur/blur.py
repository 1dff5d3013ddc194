"""2-D orthonormal DCT-II / DCT-III over the 32x32 plane of NHWC batches (blur_jax/blur.py:99-106).

The reference builds the DCT from a complex FFT with even/odd reordering and twiddles (blur.py:11-85 over
the vendored fft.py); mathematically that is the orthonormal DCT-II, which the HIP kernel evaluates
as two LDS-resident 32x32 matrix products per plane."""
from .. import ops


def batch_img_dct(xs):
    return ops.dct2d_32(xs, inverse=False)


def batch_img_idct(ys):
    return ops.dct2d_32(ys, inverse=True)
