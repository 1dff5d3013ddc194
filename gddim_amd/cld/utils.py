"""Tiny helpers of cld_jax/utils.py:172-185: the sample shape of a config and the two 2x2 einsums (HIP kernel behind both)."""
from ..ops import bmm, sbmm  # noqa: F401  einsum('bij,b...dj->b...di') / einsum('ij,b...dj->b...di')


def get_data_shape(config):
    """(H, W, C) of one sample; the 2-D point-set datasets ("ps" in their name) are flat vectors of config.data.dim."""
    data = config.data
    side = data.image_size
    return (data.dim,) if "ps" in data.dataset else (side, side, data.num_channels)
