"""gd_conv3x3_winograd2d_split6 (the operator of precision mode 2): F(2x2,3x3) with the component GEMMs in the exact three-piece bf16 split on
the bf16 matrix core, every output against a float64 convolution, next to the production fp32 kernels on the same inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _conv64(x, w_packed, bias, res, out_scale):
    B, H, W, C = x.shape
    N = w_packed.shape[0]
    w = w_packed.double().reshape(N, 3, 3, C).permute(0, 3, 1, 2)
    y = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w, bias=None if bias is None else bias.double(), padding=1).permute(0, 2, 3, 1)
    if res is not None:
        y = y + res.double()
    return y * out_scale


def _case(B, H, W, C, N, seed, with_res=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, H, W, C), generator=g)
    x = x * torch.sigmoid(x)  # swish(N(0,1)): what the network feeds its convolutions
    w = torch.randn((N, 9 * C), generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn((N,), generator=g) * 0.1
    r = torch.randn((B, H, W, N), generator=g) if with_res else None
    return x, w, b, r


@pytest.mark.parametrize("B,H,W,C,N", [(2, 16, 16, 16, 64), (3, 16, 16, 32, 64), (2, 32, 32, 48, 128), (1, 16, 32, 64, 192), (5, 32, 16, 128, 128)])
def test_split6_matches_float64(native_lib, B, H, W, C, N):
    from gddim_amd import ops
    x, w, b, r = _case(B, H, W, C, N, seed=B * 1000 + C)
    want = _conv64(x, w, b, r, 0.7)
    u = ops.conv3x3_winograd2d_split6_weights(w.cuda())
    got = ops.conv3x3_winograd2d_split6(x.cuda(), u, N, bias=b.cuda(), res=r.cuda(), out_scale=0.7).cpu().double()
    direct = ops.conv2d(x.cuda(), w.cuda(), bias=b.cuda(), res=r.cuda(), out_scale=0.7).cpu().double()
    sc = float(want.abs().max())
    err = float((got - want).abs().max())
    rms = float(((got - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt())
    msg = "split6 B=%d %dx%d %d->%d: max|e| %.3e (scale %.2f), rms rel %.3e" % (B, H, W, C, N, err, sc, rms)
    if direct is not None:
        ed = float((direct - want).abs().max())
        msg += "; direct fp32 kernel max|e| %.3e" % ed
    print(msg)
    assert torch.isfinite(got).all()
    assert err <= 4e-5 * max(1.0, sc), msg
    # no bias / residual / scale
    got2 = ops.conv3x3_winograd2d_split6(x.cuda(), u, N).cpu().double()
    assert float((got2 - _conv64(x, w, None, None, 1.0)).abs().max()) <= 4e-5 * max(1.0, sc)


def test_split6_error_is_fp32_class_at_production_shapes(native_lib):
    """The gate's criterion 1: rms error <= the production hybrid fp32 kernel's on 16x16 256->256 and 32x32 128->128
    (a few images: the error does not depend on the batch), every output against float64."""
    from gddim_amd import ops
    for H, C in ((16, 256), (32, 128)):
        x, w, b, r = _case(4, H, H, C, C, seed=H, with_res=False)
        want = _conv64(x, w, b, None, 1.0)
        u = ops.conv3x3_winograd2d_split6_weights(w.cuda())
        got = ops.conv3x3_winograd2d_split6(x.cuda(), u, C, bias=b.cuda()).cpu().double()
        hyb = ops.conv3x3_winograd42(x.cuda(), w.cuda(), bias=b.cuda()).cpu().double()
        rms = lambda a: float(((a - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt())  # noqa: E731
        e6, eh = rms(got), rms(hyb)
        print("%dx%d %d->%d: rms rel error split6 %.3e, hybrid fp32 %.3e; max|e| %.3e vs %.3e" % (H, H, C, C, e6, eh, float((got - want).abs().max()),
                                                                                                 float((hyb - want).abs().max())))
        assert e6 <= eh * 1.05, (e6, eh)


def test_split6_refuses_other_shapes(native_lib):
    from gddim_amd import ops
    from gddim_amd._abi import GddimError
    x, w, b, r = _case(1, 8, 8, 16, 64, seed=1)
    u = ops.conv3x3_winograd2d_split6_weights(w.cuda())
    with pytest.raises(GddimError, match="not eligible"):
        ops.conv3x3_winograd2d_split6(x.cuda(), u, 64)


@pytest.mark.parametrize("B,C,N,groups,keep_raw", [(3, 64, 128, 32, True), (2, 128, 256, 32, True), (5, 32, 64, 4, False), (2, 256, 256, 32, False)])
def test_split6_epilogue_groupnorm(native_lib, B, C, N, groups, keep_raw):
    """gd_conv3x3_winograd2d_split6_gn: the GroupNorm (+ swish) that follows the convolution, in the split6 kernel's epilogue on 16x16 images
    (split6 mode's Op::egn_peer path; 4, 8 and 16 channels per group), against float64 -- bias, residual and scale in front of it."""
    from gddim_amd import ops
    x, w, b, r = _case(B, 16, 16, C, N, seed=B * 100 + N + groups)
    g = torch.Generator().manual_seed(groups)
    gamma = 1 + 0.3 * torch.randn(N, generator=g)
    beta = 0.3 * torch.randn(N, generator=g)
    u = ops.conv3x3_winograd2d_split6_weights(w.cuda())
    raw, gn = ops.conv3x3_winograd2d_split6_gn(x.cuda(), u, N, gamma.cuda(), beta.cuda(), groups, act=True, bias=b.cuda(), res=r.cuda(), out_scale=0.7,
                                               keep_raw=keep_raw)
    y = _conv64(x, w, b, r, 0.7)
    yg = y.reshape(B, 256, groups, N // groups)
    mean = yg.mean(dim=(1, 3), keepdim=True)
    var = (yg * yg).mean(dim=(1, 3), keepdim=True) - mean * mean
    t = ((yg - mean) * torch.rsqrt(var + 1e-6)).reshape(B, 16, 16, N) * gamma.double() + beta.double()
    want = t * torch.sigmoid(t)
    err = float((gn.cpu().double() - want).abs().max())
    eraw = float((raw.cpu().double() - y).abs().max()) if keep_raw else 0.0
    print("split6 + GroupNorm B=%d %d->%d groups %d: max|d| %.2e (raw %.2e)" % (B, C, N, groups, err, eraw))
    assert (raw is None) == (not keep_raw)
    assert err < 2e-5 and eraw < 2e-5
    # no activation, no residual
    raw2, gn2 = ops.conv3x3_winograd2d_split6_gn(x.cuda(), u, N, gamma.cuda(), beta.cuda(), groups, act=False, bias=b.cuda())
    y2 = _conv64(x, w, b, None, 1.0)
    yg = y2.reshape(B, 256, groups, N // groups)
    mean = yg.mean(dim=(1, 3), keepdim=True)
    var = (yg * yg).mean(dim=(1, 3), keepdim=True) - mean * mean
    t2 = ((yg - mean) * torch.rsqrt(var + 1e-6)).reshape(B, 16, 16, N) * gamma.double() + beta.double()
    assert float((gn2.cpu().double() - t2).abs().max()) < 2e-5 and float((raw2.cpu().double() - y2).abs().max()) < 2e-5
