"""GPU parity tests of the stand-alone HIP operators (through the C ABI) against the CPU oracle /
plain torch-CPU float64 references on the same seeded inputs.  Tolerances are fp32-roundoff class and
are stated per test."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(native_lib):
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    assert native_lib.gd_device_count() >= 1
    return torch.device("cuda", 0)


def _rel_err(got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    return np.abs(got - want).max() / max(1e-30, np.abs(want).max())


# ---- update rules -------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_hist", [0, 1, 3])
def test_ab_step_matches_oracle(dev, n_hist):
    from gddim_amd import ops
    from oracle import cld as ocld
    rng = np.random.default_rng(n_hist)
    shape = (5, 8, 8, 3, 2)
    x = rng.standard_normal(shape).astype(np.float32)
    new_eps = rng.standard_normal(shape).astype(np.float32)
    hist = rng.standard_normal((n_hist,) + shape).astype(np.float32)
    coef = (rng.standard_normal((n_hist + 2, 2, 2)) * 3).astype(np.float32)
    want_x, want_h = ocld.multistep_ab_step(x.astype(np.float64), coef.astype(np.float64), new_eps.astype(np.float64),
                                            hist.astype(np.float64))
    got_x, got_h = ops.ab_step(torch.as_tensor(x, device=dev), coef, torch.as_tensor(new_eps, device=dev),
                               torch.as_tensor(hist, device=dev))
    assert _rel_err(got_x.cpu().numpy(), want_x) < 2e-6
    assert np.array_equal(got_h.cpu().numpy(), want_h.astype(np.float32))  # pure data movement: bit exact


def test_ab_step_scalar_and_bmm(dev):
    from gddim_amd import ops
    from oracle import blur as oblur, cld as ocld
    rng = np.random.default_rng(7)
    x = rng.standard_normal((4, 32, 32, 3)).astype(np.float32)
    ne = rng.standard_normal(x.shape).astype(np.float32)
    hist = rng.standard_normal((2,) + x.shape).astype(np.float32)
    coef = rng.standard_normal(4).astype(np.float32)
    wx, wh = oblur.ms_ab_step(x.astype(np.float64), coef.astype(np.float64), ne.astype(np.float64), hist.astype(np.float64))
    gx, gh = ops.ab_step_scalar(torch.as_tensor(x, device=dev), coef, torch.as_tensor(ne, device=dev), torch.as_tensor(hist, device=dev))
    assert _rel_err(gx.cpu().numpy(), wx) < 2e-6
    assert np.array_equal(gh.cpu().numpy(), wh.astype(np.float32))
    u = rng.standard_normal((3, 4, 4, 3, 2)).astype(np.float32)
    m = rng.standard_normal((2, 2)).astype(np.float32)
    mb = rng.standard_normal((3, 2, 2)).astype(np.float32)
    assert _rel_err(ops.sbmm(m, torch.as_tensor(u, device=dev)).cpu().numpy(), ocld.sbmm(m.astype(np.float64), u.astype(np.float64))) < 1e-6
    assert _rel_err(ops.bmm(mb, torch.as_tensor(u, device=dev)).cpu().numpy(), ocld.bmm(mb.astype(np.float64), u.astype(np.float64))) < 1e-6


def test_dct_matches_oracle_and_scipy(dev):
    import scipy.fft
    from gddim_amd import ops
    from oracle import blur as oblur
    rng = np.random.default_rng(3)
    x = rng.standard_normal((3, 32, 32, 3)).astype(np.float32)
    y = ops.dct2d_32(torch.as_tensor(x, device=dev)).cpu().numpy()
    want = oblur.batch_img_dct(x.astype(np.float64))
    assert np.abs(y - want).max() < 2e-5
    assert np.abs(want - scipy.fft.dctn(x.astype(np.float64), axes=(1, 2), norm="ortho")).max() < 1e-12
    back = ops.dct2d_32(torch.as_tensor(y, device=dev), inverse=True).cpu().numpy()
    assert np.abs(back - x).max() < 2e-5  # round trip
    assert np.abs(back - oblur.batch_img_idct(y.astype(np.float64))).max() < 2e-5


# ---- implicit-GEMM convolution ----------------------------------------------------------------------------
def _conv_case(dev, B, H, W, Cin, Cout, k, stride, pad, bias, res, seed, alpha=1.0, out_scale=1.0):
    from gddim_amd import ops
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, Cin, Cout)) / math.sqrt(k * k * Cin)).astype(np.float32)  # HWIO
    b = rng.standard_normal(Cout).astype(np.float32) if bias else None
    want = F.conv2d(torch.as_tensor(x, dtype=torch.float64).permute(0, 3, 1, 2),
                    torch.as_tensor(w, dtype=torch.float64).permute(3, 2, 0, 1), stride=stride, padding=pad).permute(0, 2, 3, 1)
    want = alpha * want
    if bias:
        want = want + torch.as_tensor(b, dtype=torch.float64)
    r = None
    if res:
        r = rng.standard_normal(tuple(want.shape)).astype(np.float32)
        want = want + torch.as_tensor(r, dtype=torch.float64)
    want = (want * out_scale).numpy()
    wp = np.ascontiguousarray(np.transpose(w, (3, 0, 1, 2))).reshape(Cout, k * k * Cin)  # [O][kh][kw][I]
    got = ops.conv2d(torch.as_tensor(x, device=dev), torch.as_tensor(wp, device=dev),
                     None if b is None else torch.as_tensor(b, device=dev), None if r is None else torch.as_tensor(r, device=dev),
                     KH=k, KW=k, stride=stride, pad=pad, alpha=alpha, out_scale=out_scale).cpu().numpy()
    assert got.shape == want.shape
    return np.abs(got - want).max() / max(1.0, np.abs(want).max())


@pytest.mark.parametrize("case", [
    # B, H, W, Cin, Cout, k, stride, pad, bias, res   -- the U-Net's layer shapes at small batch
    (2, 8, 8, 128, 128, 3, 1, 1, True, False),       # 64x64 tiles
    (3, 16, 16, 256, 256, 3, 1, 1, True, True),      # residual epilogue
    (2, 32, 32, 8, 128, 3, 1, 1, True, False),       # stem: K = 72 (not a multiple of 32)
    (2, 32, 32, 128, 6, 3, 1, 1, True, False),       # head: N = 6 (masked columns)
    (2, 16, 16, 384, 256, 1, 1, 0, True, False),     # 1x1 shortcut
    (2, 32, 32, 8, 128, 6, 2, 2, True, True),        # pyramid 6x6 stride 2 pad 2
    (3, 16, 16, 128, 256, 6, 2, 2, True, True),      # pyramid 6x6 stride 2 with Cin % 32 == 0: the 36-tap (two mask words) fast path
    (2, 8, 10, 64, 64, 6, 2, 2, False, False),       # the same on a non-square image, borders on all sides
    (5, 4, 4, 512, 256, 3, 1, 1, False, False),      # rows of several images in one tile, M = 80 (ragged)
    (1, 1, 1, 256, 512, 1, 1, 0, True, False),       # dense layer, M = 1
    (48, 32, 32, 128, 128, 3, 1, 1, True, True),     # 49152 rows -> 128x128 tiles
    (96, 16, 16, 64, 64, 3, 1, 1, True, False),      # 24576 rows, N = 64 -> 128x64 tiles
    (256, 4, 4, 256, 256, 3, 1, 1, True, True),      # the 4x4 level at the production batch: 256 tiles, K = 2304 -> split-K x4 in the workgroup
    (4, 4, 4, 512, 256, 1, 1, 0, True, True),        # K = 512 (16 chunks) on few tiles -> split-K x2
    (3, 4, 4, 288, 64, 3, 1, 1, True, False),        # K = 2592 = 81 chunks: not a multiple of the split (zero-filled tail chunks)
    (65, 32, 32, 256, 128, 1, 1, 0, True, True),     # 520 tiles of K = 256: the persistent short-K kernel, two tiles per workgroup + residual
    (70, 30, 30, 128, 256, 1, 1, 0, True, False),    # persistent kernel: ragged last M tile (63000 rows), two N tiles, K = 128 (4 chunks)
    (70, 32, 32, 96, 128, 3, 1, 1, True, True),      # persistent kernel on a 3x3 window (K = 864, 27 chunks: borders through the tap masks)
])
def test_conv2d_matches_torch_fp64(dev, case):
    err = _conv_case(dev, *case, seed=hash(case) % 1000, alpha=1.0, out_scale=1.0 / math.sqrt(2.0) if case[-1] else 1.0)
    assert err < 2e-5, (case, err)  # fp32 MFMA accumulate over K <= 4608


def test_conv2d_transpose_detecting(dev):
    """A = identity-like input with an asymmetric kernel: catches swapped row/col or tap order."""
    from gddim_amd import ops
    B, H, W, Cin, Cout = 1, 8, 8, 4, 36
    x = torch.zeros(B, H, W, Cin)
    x[0, 3, 5, 2] = 1.0  # a single impulse
    w = torch.arange(3 * 3 * Cin * Cout, dtype=torch.float32).reshape(3, 3, Cin, Cout) / 100.0
    wp = w.permute(3, 0, 1, 2).reshape(Cout, 9 * Cin).contiguous()
    got = ops.conv2d(x.to(dev), wp.to(dev), KH=3, KW=3, stride=1, pad=1).cpu()
    want = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(), padding=1).permute(0, 2, 3, 1)
    assert (got.double() - want).abs().max() < 1e-5


# ---- fp32 Winograd kernels: F(2x2,3x3) (round 2; the 4x4 level and the A/B fallback) and the hybrid F(2,3)y x F(4,3)x (production) ----
WINO2D_CASES = [
    # B, H, W, Cin, Cout, bias, res
    (2, 32, 32, 128, 128, True, True),    # the dominant layer shape: 4 tiles per image, 2 N tiles
    (3, 16, 16, 256, 256, True, True),    # W = 16: a 16-quad unit is two quad rows
    (1, 64, 64, 32, 68, False, True),     # W = 64: a unit is half a quad row; ragged N tile (68 = 64 + 4); K of one slab
    (5, 16, 32, 96, 132, True, True),     # non-square image (128 quads per image), Cin = 3 slabs, ragged N
    (3, 8, 16, 64, 64, False, False),     # 32 quads per image: the last tile holds one image only (ragged M: 96 quads)
    (2, 24, 32, 32, 32, True, False),     # H * W / 4 = 192 quads per image (not a power of two), N = 32
    (6, 8, 8, 64, 96, True, True),        # W = 8: a 16-quad unit is a whole 8x8 image (four quad rows); ragged M (96 quads), ragged N
    (3, 16, 8, 32, 64, False, True),      # W = 8, H = 16: two units per image
    (5, 4, 16, 32, 64, True, True),       # 16 quads per image: every unit is a whole (4 x 16) image
    (2, 32, 32, 64, 6, True, False),      # N = 6: the U-Net's head convolution (one N tile, 58 of its 64 columns past the end)
    (2, 16, 16, 32, 18, False, True),     # N = 18: not a multiple of 4, residual row stride 18
    (64, 4, 4, 256, 256, True, True),     # 4x4 images: the channel range split over 8 workgroups per tile + the finishing pass
    (7, 4, 4, 64, 36, True, True),        # ... 28 quads (a ragged tile, a unit that is half past the batch), two splits of four chunks, ragged N
    (256, 4, 4, 128, 64, False, False),   # ... 16 tiles x 1 N tile, four splits of four chunks
    (1024, 4, 4, 256, 256, True, True),   # ... B = 1024 (run_lib's eval.batch_size on one GPU): 256 tiles = one per CU already; two splits walked in rounds (round 5)
]


@pytest.mark.parametrize("case", WINO2D_CASES)
def test_conv3x3_winograd2d_matches_torch_fp64(dev, case):
    """gd_conv3x3_winograd2d (F(2x2,3x3): 16 products per 2x2 output quad) against the float64 direct convolution; same bar as the
    direct kernel: both transforms are additions, the 1/2 and 1/4 of the weight transform are exact."""
    from gddim_amd import ops
    B, H, W, Cin, Cout, use_bias, use_res = case
    g = torch.Generator().manual_seed(sum(case[:5]) + 1)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(3, 3, Cin, Cout, generator=g) / math.sqrt(9 * Cin)
    bias = torch.randn(Cout, generator=g) if use_bias else None
    res = torch.randn(B, H, W, Cout, generator=g) if use_res else None
    wp = w.permute(3, 0, 1, 2).reshape(Cout, 9 * Cin).contiguous()
    got = ops.conv3x3_winograd2d(x.to(dev), wp.to(dev), None if bias is None else bias.to(dev), None if res is None else res.to(dev),
                                 alpha=1.25, out_scale=0.7).cpu().double()
    want = 1.25 * F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(), padding=1).permute(0, 2, 3, 1)
    if bias is not None:
        want = want + bias.double()
    if res is not None:
        want = want + res.double()
    want = want * 0.7
    direct = ops.conv2d(x.to(dev), wp.to(dev), None if bias is None else bias.to(dev), None if res is None else res.to(dev),
                        alpha=1.25, out_scale=0.7).cpu().double()
    err, err_direct = (got - want).abs().max().item(), (direct - want).abs().max().item()
    print("wino2d", case, "err %.2e direct %.2e" % (err, err_direct))
    assert err < 2e-5 and err < 4 * err_direct + 1e-6, (case, err, err_direct)


def test_conv3x3_winograd2d_impulse_and_rejects(dev):
    """Impulses with an asymmetric kernel at corners, edges and odd / even positions (catches a swapped tap, parity or component
    order in either direction); shapes the kernel cannot take are refused loudly."""
    from gddim_amd import ops
    from gddim_amd._abi import GddimError
    B, H, W, Cin, Cout = 1, 16, 16, 32, 64
    w = torch.arange(3 * 3 * Cin * Cout, dtype=torch.float32).reshape(3, 3, Cin, Cout) / 1000.0
    wp = w.permute(3, 0, 1, 2).reshape(Cout, 9 * Cin).contiguous()
    for (y, xx, c) in [(0, 0, 5), (3, 5, 17), (15, 15, 31), (4, 2, 0), (15, 0, 9), (0, 15, 3), (7, 8, 12), (8, 7, 1)]:
        x = torch.zeros(B, H, W, Cin)
        x[0, y, xx, c] = 1.0
        got = ops.conv3x3_winograd2d(x.to(dev), wp.to(dev)).cpu().double()
        want = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(), padding=1).permute(0, 2, 3, 1)
        assert (got - want).abs().max() < 1e-4 * want.abs().max(), (y, xx, c)
    for shape in [(1, 4, 4, 32, 64), (1, 4, 4, 64, 30), (1, 16, 16, 24, 64), (1, 15, 16, 32, 64), (1, 2, 16, 32, 64), (1, 12, 8, 32, 64)]:
        b_, h_, w_, ci, co = shape
        with pytest.raises(GddimError):
            ops.conv3x3_winograd2d(torch.zeros(b_, h_, w_, ci, device=dev), torch.zeros(co, 9 * ci, device=dev))


WINO42_CASES = [
    # B, H, W, Cin, Cout, bias, res
    (2, 32, 32, 128, 128, True, True),    # the dominant layer shape: 4 M blocks per image (8 tiles per tile row), 2 N tiles
    (3, 16, 16, 256, 256, True, True),    # W = 16: an 8-tile unit is two tile rows; one image per M block
    (1, 64, 64, 32, 68, False, True),     # W = 64: a unit is half a tile row; ragged N tile (68 = 64 + 4); K of one slab
    (5, 16, 32, 96, 132, True, True),     # non-square image (64 tiles per image), Cin = 3 slabs, ragged N
    (3, 8, 16, 64, 64, False, False),     # 16 tiles per image: the last M block holds one image only (ragged M: 48 tiles)
    (2, 24, 32, 32, 32, True, False),     # 96 tiles per image (not a power of two), N = 32: the second output block of every wave past the end
    (6, 8, 8, 64, 96, True, True),        # W = 8: a unit is a whole 8x8 image (four tile rows of two); ragged M (48 tiles), ragged N
    (3, 16, 8, 32, 64, False, True),      # W = 8, H = 16: two units per image
    (5, 4, 16, 32, 64, True, True),       # 8 tiles per image: every unit is a whole (4 x 16) image
    (4, 2, 32, 32, 64, True, True),       # one tile row per image
    (2, 32, 32, 64, 6, True, False),      # N = 6: the U-Net's head convolution (one N tile, 58 of its 64 columns past the end)
    (2, 16, 16, 32, 18, False, True),     # N = 18: not a multiple of 4, residual row stride 18
    (2, 16, 16, 512, 256, True, True),    # the longest K of the U-Net (64 chunks)
]


@pytest.mark.parametrize("case", [
    (3, 8, 8, 64, 64, 16, True, None),       # one image = one wave unit
    (3, 16, 16, 64, 128, 32, True, None),    # one image = one workgroup tile
    (5, 16, 16, 128, 128, 32, False, None),  # ... sole reader: the raw output is never stored; two N tiles
    (5, 8, 8, 64, 32, 8, True, 2),           # half an N tile, two workgroups walking the tiles of 5 images
    (6, 16, 16, 256, 128, 32, False, 1),     # two-chunk sources, ONE workgroup walking 6 images x 2 N tiles
])
@pytest.mark.parametrize("nw", [2, 1])
def test_conv3x3_winograd42_epilogue_groupnorm(dev, case, nw, gd_env):
    """gd_conv3x3_winograd42_gn: the GroupNorm (+ swish) that follows a 3x3 convolution, computed in the hybrid Winograd kernel's epilogue
    (the engine's Op::egn_peer path), against float64 -- bias, residual and scale in front of the normalisation as in a residual block.
    nw: the 64-output (eight waves) and the 32-output (four waves, round 4) workgroup tile."""
    from gddim_amd import ops
    B, H, W, Cin, Cout, groups, keep_raw, grid = case
    gd_env(GDDIM_WINO42_GRID=grid if grid else None, GDDIM_WINO42_NW=nw)
    g = torch.Generator().manual_seed(sum(case[:6]))
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(3, 3, Cin, Cout, generator=g) / math.sqrt(9 * Cin)
    bias = torch.randn(Cout, generator=g)
    res = torch.randn(B, H, W, Cout, generator=g)
    gamma = 1 + 0.3 * torch.randn(Cout, generator=g)
    beta = 0.3 * torch.randn(Cout, generator=g)
    wp = w.permute(3, 0, 1, 2).reshape(Cout, 9 * Cin).contiguous()
    raw, gn = ops.conv3x3_winograd42_gn(x.to(dev), wp.to(dev), gamma.to(dev), beta.to(dev), groups, act=True, bias=bias.to(dev), res=res.to(dev),
                                        out_scale=0.7, keep_raw=keep_raw)
    y = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    y = (y + bias.double() + res.double()) * 0.7
    yg = y.reshape(B, H * W, groups, Cout // groups)
    mean = yg.mean(dim=(1, 3), keepdim=True)
    var = (yg * yg).mean(dim=(1, 3), keepdim=True) - mean * mean
    t = ((yg - mean) * torch.rsqrt(var + 1e-6)).reshape(B, H, W, Cout) * gamma.double() + beta.double()
    want = t * torch.sigmoid(t)
    err = (gn.cpu().double() - want).abs().max().item()
    eraw = (raw.cpu().double() - y).abs().max().item() if keep_raw else 0.0
    print("wino42 + GroupNorm %s: max|d| %.2e (raw %.2e)" % (case, err, eraw))
    assert err < 5e-5 and eraw < 5e-5


@pytest.mark.parametrize("case", WINO42_CASES)
@pytest.mark.parametrize("nw", [2, 1])
def test_conv3x3_winograd42_matches_torch_fp64(dev, case, nw, gd_env):
    """gd_conv3x3_winograd42 (hybrid F(2,3)_y x F(4,3)_x: 24 products per 2 x 4 output tile) against the float64 direct convolution.
    F(4,3) has transform constants up to 8 (A^T) and 5 (B^T): the bar is 4e-5 absolute on outputs of scale ~3 and at most 6x the
    direct fp32 kernel's error (measured 1.5-2.5x; CPU emulation in profiles/r03_winograd_error_study.md).  nw: the workgroup tile --
    32 tiles x 64 outputs (eight waves, the production tile) or 32 tiles x 32 outputs (four waves staging two tiles' patch rows each, the
    tile of launches that cannot fill the chip: round 4); Cin = 32 has the wide two-chunk form only."""
    from gddim_amd import ops
    B, H, W, Cin, Cout, use_bias, use_res = case
    gd_env(GDDIM_WINO42_NW=nw)
    g = torch.Generator().manual_seed(sum(case[:5]) + 2)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(3, 3, Cin, Cout, generator=g) / math.sqrt(9 * Cin)
    bias = torch.randn(Cout, generator=g) if use_bias else None
    res = torch.randn(B, H, W, Cout, generator=g) if use_res else None
    wp = w.permute(3, 0, 1, 2).reshape(Cout, 9 * Cin).contiguous()
    got = ops.conv3x3_winograd42(x.to(dev), wp.to(dev), None if bias is None else bias.to(dev), None if res is None else res.to(dev),
                                 alpha=1.25, out_scale=0.7).cpu().double()
    want = 1.25 * F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(), padding=1).permute(0, 2, 3, 1)
    if bias is not None:
        want = want + bias.double()
    if res is not None:
        want = want + res.double()
    want = want * 0.7
    direct = ops.conv2d(x.to(dev), wp.to(dev), None if bias is None else bias.to(dev), None if res is None else res.to(dev),
                        alpha=1.25, out_scale=0.7).cpu().double()
    err, err_direct = (got - want).abs().max().item(), (direct - want).abs().max().item()
    print("wino42", case, "err %.2e direct %.2e" % (err, err_direct))
    assert err < 4e-5 and err < 6 * err_direct + 1e-6, (case, err, err_direct)


@pytest.mark.parametrize("case", [
    (256, 32, 32, 128, 128),   # the dominant launch of the benchmark: 1 024 M blocks x 2 N tiles = 2 048 workgroup tiles, 8 rounds of the persistent walk
    (256, 8, 8, 256, 256),     # 8x8 level at B = 256: 64 M blocks x 4 N tiles = 256 workgroup tiles, exactly one round on 256 CUs
    (320, 16, 16, 256, 256),   # a batch whose walk ends in a partial round (the stream-K tail's case)
    (128, 8, 8, 512, 256),     # the reference's own per-device batch on the longest K
])
def test_conv3x3_winograd42_production_size_every_output(dev, case):
    """gd_conv3x3_winograd42 at the PRODUCTION launch sizes (default grid, default tile choice, nothing forced), every output against the
    float64 direct convolution: the persistent walk over thousands of tiles, the XCD-grouped tile ids and the far end of the
    batch, which the small operator cases above only reach through forced grids.  Besides the global bar the
    error is reported per image so that a defect confined to mid-batch positions cannot hide behind a max over the batch."""
    from gddim_amd import ops
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case) + 11)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(3, 3, Cin, Cout, generator=g) / math.sqrt(9 * Cin)
    bias = torch.randn(Cout, generator=g)
    res = torch.randn(B, H, W, Cout, generator=g)
    wp = w.permute(3, 0, 1, 2).reshape(Cout, 9 * Cin).contiguous()
    got = ops.conv3x3_winograd42(x.to(dev), wp.to(dev), bias.to(dev), res.to(dev), alpha=1.0, out_scale=0.7).cpu()
    assert torch.isfinite(got).all()
    wd = w.permute(3, 2, 0, 1).double()
    per_image = torch.empty(B, dtype=torch.float64)
    for b0 in range(0, B, 32):   # float64 reference in slabs of 32 images (memory)
        xs = x[b0:b0 + 32].permute(0, 3, 1, 2).double()
        want = (F.conv2d(xs, wd, padding=1).permute(0, 2, 3, 1) + bias.double() + res[b0:b0 + 32].double()) * 0.7
        per_image[b0:b0 + 32] = (got[b0:b0 + 32].double() - want).abs().amax(dim=(1, 2, 3))
    print("wino42 production size", case, "max|d| %.2e, worst image %d, median over images %.2e" %
          (per_image.max().item(), int(per_image.argmax()), per_image.median().item()))
    assert per_image.max().item() < 4e-5, (case, int(per_image.argmax()), per_image.max().item())
    assert per_image.max().item() < 4 * per_image.median().item() + 1e-6  # no image is an outlier: the error is round-off everywhere


@pytest.mark.parametrize("grid", [1, 3, 8])
@pytest.mark.parametrize("case", [(2, 32, 32, 128, 128, True, True), (5, 32, 32, 64, 132, True, True), (6, 8, 8, 64, 96, True, False),
                                  (3, 16, 16, 256, 256, False, True)])
@pytest.mark.parametrize("nw", [2, 1])
def test_conv3x3_winograd42_persistent_walk(dev, case, grid, nw, gd_env):
    """The hybrid kernel is persistent (Cin >= 64): a workgroup walks over tiles and its K-loop pipeline runs across them (the last two
    chunks of a tile stage the first two of the next).  Unit-test problems have fewer tiles than CUs, so GDDIM_WINO42_GRID forces a small
    grid: every workgroup then takes several tiles -- including the XCD-grouped mapping's padding ids (20 M tiles: ids 20..23 of the last
    group of 8 are skipped), ragged last tiles and a walk that ends on different tiles per workgroup.  Result must equal the default
    launch bit for bit (same arithmetic, another schedule) and match float64."""
    import os
    from gddim_amd import ops
    B, H, W, Cin, Cout, use_bias, use_res = case
    g = torch.Generator().manual_seed(sum(case[:5]) + 3)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(3, 3, Cin, Cout, generator=g) / math.sqrt(9 * Cin)
    bias = torch.randn(Cout, generator=g) if use_bias else None
    res = torch.randn(B, H, W, Cout, generator=g) if use_res else None
    wp = w.permute(3, 0, 1, 2).reshape(Cout, 9 * Cin).contiguous()
    args = (x.to(dev), wp.to(dev), None if bias is None else bias.to(dev), None if res is None else res.to(dev))
    gd_env(GDDIM_WINO42_NW=2)
    ref = ops.conv3x3_winograd42(*args, alpha=1.25, out_scale=0.7).cpu()  # the wide tile, default grid: every output's arithmetic is the same in both tiles
    gd_env(GDDIM_WINO42_GRID=grid, GDDIM_WINO42_NW=nw)
    got = ops.conv3x3_winograd42(*args, alpha=1.25, out_scale=0.7).cpu()
    assert torch.equal(got, ref), (case, grid, (got - ref).abs().max().item())
    want = 1.25 * F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(), padding=1).permute(0, 2, 3, 1)
    if bias is not None:
        want = want + bias.double()
    if res is not None:
        want = want + res.double()
    assert (got.double() - want * 0.7).abs().max().item() < 4e-5


def test_conv3x3_winograd42_impulse_and_rejects(dev):
    """Impulses with an asymmetric kernel at corners, edges and every position modulo the 2 x 4 tile (catches a swapped tap, a
    component or an output-column order in either direction); shapes the kernel cannot take are refused loudly."""
    from gddim_amd import ops
    from gddim_amd._abi import GddimError
    B, H, W, Cin, Cout = 1, 16, 16, 32, 64
    w = torch.arange(3 * 3 * Cin * Cout, dtype=torch.float32).reshape(3, 3, Cin, Cout) / 1000.0
    wp = w.permute(3, 0, 1, 2).reshape(Cout, 9 * Cin).contiguous()
    for (y, xx, c) in [(0, 0, 5), (3, 5, 17), (15, 15, 31), (4, 2, 0), (15, 0, 9), (0, 15, 3), (7, 8, 12), (8, 7, 1), (5, 3, 2), (6, 12, 30),
                       (9, 4, 7), (10, 13, 8), (1, 6, 11)]:
        x = torch.zeros(B, H, W, Cin)
        x[0, y, xx, c] = 1.0
        got = ops.conv3x3_winograd42(x.to(dev), wp.to(dev)).cpu().double()
        want = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(), padding=1).permute(0, 2, 3, 1)
        assert (got - want).abs().max() < 1e-4 * want.abs().max(), (y, xx, c)
    for shape in [(1, 4, 4, 32, 64), (1, 16, 16, 24, 64), (1, 15, 16, 32, 64), (1, 2, 16, 32, 64), (1, 12, 8, 32, 64), (1, 16, 12, 32, 64)]:
        b_, h_, w_, ci, co = shape
        with pytest.raises(GddimError):
            ops.conv3x3_winograd42(torch.zeros(b_, h_, w_, ci, device=dev), torch.zeros(co, 9 * ci, device=dev))


# ---- GroupNorm / FIR / attention -----------------------------------------------------------------------------
@pytest.mark.parametrize("B,HW,C,act", [(3, 1024, 128, 1), (2, 256, 256, 1), (2, 64, 384, 1), (5, 16, 512, 0), (2, 256, 16, 1),
                                        (2, 64, 192, 1)])  # 192 channels: 32 groups of 6 -> the generic path
def test_group_norm_matches_oracle(dev, B, HW, C, act):
    from gddim_amd import ops
    from oracle import ncsnpp as onet
    rng = np.random.default_rng(C + HW)
    side = int(math.isqrt(HW))
    x = (rng.standard_normal((B, side, side, C)) * 2 + 0.5).astype(np.float32)
    p = {"scale": (1 + 0.2 * rng.standard_normal(C)).astype(np.float32), "bias": (0.3 * rng.standard_normal(C)).astype(np.float32)}
    want = onet.group_norm(torch.as_tensor(x, dtype=torch.float64), p)
    if act:
        want = onet.swish(want)
    G = min(C // 4, 32)
    got = ops.group_norm(torch.as_tensor(x, device=dev), torch.as_tensor(p["scale"], device=dev), torch.as_tensor(p["bias"], device=dev),
                         G, act=bool(act)).cpu().numpy()
    assert np.abs(got - want.numpy()).max() < 3e-5


@pytest.mark.parametrize("up", [True, False])
def test_fir_resample_matches_oracle(dev, up):
    from gddim_amd import ops
    from oracle import ncsnpp as onet
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 8, 8, 16)).astype(np.float32)
    xt = torch.as_tensor(x, dtype=torch.float64)
    want = onet.upsample_2d(xt, [1, 3, 3, 1]) if up else onet.downsample_2d(xt, [1, 3, 3, 1])
    got = ops.fir_resample(torch.as_tensor(x, device=dev), up).cpu().numpy()
    assert got.shape == tuple(want.shape)
    assert np.abs(got - want.numpy()).max() < 2e-6


@pytest.mark.parametrize("B,N,C", [(3, 256, 256), (4, 16, 256), (2, 64, 32), (160, 256, 256)])  # 160 x 4 tiles: the persistent kernel over z slices
def test_attention_matches_torch(dev, B, N, C):
    from gddim_amd import ops
    rng = np.random.default_rng(N)
    qkv = rng.standard_normal((B, N, 3 * C)).astype(np.float32)
    q, k, v = (torch.as_tensor(qkv[..., i * C:(i + 1) * C], dtype=torch.float64) for i in range(3))
    w = torch.softmax(torch.einsum("bnc,bmc->bnm", q, k) * (C ** -0.5), dim=-1)
    want = torch.einsum("bnm,bmc->bnc", w, v).numpy()
    got = ops.attention(torch.as_tensor(qkv, device=dev)).cpu().numpy()
    assert np.abs(got - want).max() < 2e-5


def test_ops_refuse_cpu_tensors(native_lib):
    """No CPU fallback: the operator API fails loudly on host tensors."""
    from gddim_amd import ops
    from gddim_amd._abi import GddimError
    with pytest.raises(GddimError):
        ops.sbmm(np.eye(2, dtype=np.float32), torch.zeros(1, 2, 2))


# ---- shape fuzzing of the implicit-GEMM entry point (hypothesis) -----------------------------------------------------------------
def test_conv2d_shape_fuzz(dev):
    """Random shapes across both kernel paths (fast: Cin % 32 == 0; generic: anything % 4 == 0), ragged M / N tiles, strides and
    paddings, against torch float64 through gd_conv2d."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    @settings(max_examples=30, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(B=st.integers(1, 3), H=st.integers(3, 20), W=st.integers(3, 20), cin4=st.integers(1, 24), Cout=st.integers(1, 150),
           k=st.sampled_from([1, 3]), stride=st.sampled_from([1, 2]), bias=st.booleans(), res=st.booleans(),
           fast=st.booleans())
    def run(B, H, W, cin4, Cout, k, stride, bias, res, fast):
        Cin = (cin4 % 6 + 1) * 32 if fast else cin4 * 4
        pad = k // 2
        err = _conv_case(dev, B, H, W, Cin, Cout, k, stride, pad, bias, res, seed=B * 7 + H * 11 + W * 13 + Cin + Cout,
                         alpha=1.0, out_scale=0.5 if res else 1.0)
        assert err < 2e-5, (B, H, W, Cin, Cout, k, stride, bias, res, err)

    run()


@pytest.mark.parametrize("up", [True, False])
def test_naive_resample_matches_oracle(dev, up):
    from gddim_amd import ops
    from oracle import ncsnpp as onet
    rng = np.random.default_rng(12)
    x = rng.standard_normal((3, 8, 6, 20)).astype(np.float32)
    xt = torch.as_tensor(x, dtype=torch.float64)
    want = onet.naive_upsample_2d(xt) if up else onet.naive_downsample_2d(xt)
    got = ops.fir_resample(torch.as_tensor(x, device=dev), up, naive=True).cpu().numpy()
    assert got.shape == tuple(want.shape) and np.abs(got - want.numpy()).max() < 1e-6
    if up:
        assert np.array_equal(got, np.repeat(np.repeat(x, 2, axis=1), 2, axis=2))  # exact: pure replication


@pytest.mark.parametrize("qblock", [64, 32])
def test_attention_fused_matches_float64(dev, qblock, gd_env):
    """gd_attention_fused (q k^T -> softmax -> p v in one launch, N = C = 256, operands in the engine's layout: q|k packed, v transposed per
    image, value bias in the epilogue) against float64 and against the three-launch path; scores of very different magnitude per row
    (the row maximum matters), other sizes refused.  qblock: the 64-query form (one workgroup per CU) and the 32-query form (two per CU,
    round 4: what the launcher takes when 4 B workgroups cannot fill the chip)."""
    from gddim_amd import ops
    from gddim_amd._abi import GddimError
    gd_env(GDDIM_ATTN_Q=qblock)
    B, N, C = 3, 256, 256
    g = torch.Generator().manual_seed(7)
    q = torch.randn(B, N, C, generator=g) * torch.linspace(0.2, 6.0, N).view(1, N, 1)  # rows with flat and with peaked distributions
    k = torch.randn(B, N, C, generator=g)
    v = torch.randn(B, N, C, generator=g)
    bias = torch.randn(C, generator=g)
    qk = torch.cat([q, k], dim=-1).contiguous()
    vT = v.transpose(1, 2).contiguous()
    got = ops.attention_fused(qk.to(dev), vT.to(dev), bias.to(dev)).cpu().double()
    w = torch.softmax(torch.einsum("bic,bjc->bij", q.double(), k.double()) * C ** -0.5, dim=-1)
    want = torch.einsum("bij,bjc->bic", w, v.double()) + bias.double()
    three = ops.attention(torch.cat([q, k, v], dim=-1).contiguous().to(dev)).cpu().double() + bias.double()
    err, err3 = (got - want).abs().max().item(), (three - want).abs().max().item()
    print("fused attention: max|d| %.2e (three launches %.2e), |out| %.2f" % (err, err3, want.abs().max().item()))
    assert err < 2e-5 and err < 4 * err3 + 1e-6
    with pytest.raises(GddimError):
        ops.attention_fused(torch.zeros(1, 64, 512, device=dev), torch.zeros(1, 256, 64, device=dev))


@pytest.mark.parametrize("qblock", [64, 32])
def test_attention_block_fused_epilogue(dev, qblock, gd_env):
    """gd_attention_block_fused: the fused core with separate q / k tensors and the block tail in its epilogue -- bias, residual,
    1/sqrt(2) and the GroupNorm quad statistics of the result (kernels/qstats.h layout) -- against float64; a batch that is not a
    multiple of the 8 images one XCD round holds.  Both query blocks; the two forms agree bit for bit (same k order in every row)."""
    from gddim_amd import ops
    gd_env(GDDIM_ATTN_Q=qblock)
    B, N, C = 11, 256, 256
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, N, C, generator=g) * 1.5
    k = torch.randn(B, N, C, generator=g)
    v = torch.randn(B, N, C, generator=g)
    bias = torch.randn(C, generator=g)
    res = torch.randn(B, N, C, generator=g) * 2
    sc = 2 ** -0.5
    out, qs = ops.attention_block_fused(q.to(dev), k.to(dev), v.transpose(1, 2).contiguous().to(dev), bias.to(dev), res.to(dev), sc, want_stats=True)
    w = torch.softmax(torch.einsum("bic,bjc->bij", q.double(), k.double()) * C ** -0.5, dim=-1)
    want = (res.double() + torch.einsum("bij,bjc->bic", w, v.double()) + bias.double()) * sc
    err = (out.cpu().double() - want).abs().max().item()
    wv = want.reshape(B, N // 32, 32, C // 4, 4)
    ws, wq = wv.sum(dim=(2, 4)), (wv * wv).sum(dim=(2, 4))
    es = (qs[..., 0].cpu().double() - ws).abs().max().item()
    eq = ((qs[..., 1].cpu().double() - wq).abs() / wq).max().item()
    print("attention block epilogue: max|d| %.2e, statistics |d sum| %.2e, rel d sumsq %.2e" % (err, es, eq))
    assert err < 2e-5 and es < 2e-3 and eq < 1e-5
    plain = ops.attention_block_fused(q.to(dev), k.to(dev), v.transpose(1, 2).contiguous().to(dev))  # no bias, residual, statistics
    assert (plain.cpu().double() - torch.einsum("bij,bjc->bic", w, v.double())).abs().max().item() < 2e-5
    gd_env(GDDIM_ATTN_Q=96 - qblock)
    other, qs2 = ops.attention_block_fused(q.to(dev), k.to(dev), v.transpose(1, 2).contiguous().to(dev), bias.to(dev), res.to(dev), sc, want_stats=True)
    assert torch.equal(other, out) and torch.equal(qs2, qs)
