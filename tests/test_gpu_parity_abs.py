"""Full-length parity at the north-star's own bar: |x_hip - x_oracle| <= 1e-3 per pixel, ABSOLUTE, images in [0, 1] units.

The other sampler tests use random weights, for which nothing cancels the reverse-time transition (|x| grows to 1e3..1e5) and the
bar has to be read relative to the output scale.  Here the score model is a *near-denoiser*, so the trajectory stays O(1) over all
NFE steps and the absolute bar applies, on the production configuration (BASELINE configs 2, 3, 5: full accr / ddpm_deep
NCSN++, 256 images per call, NFE 50 / 100, hipGraph replay, the production kernels):

  CLD   eps(u, t) = NCSNpp(u, t) + X(t) u with X(t) = R(t)^T S(t)^-1, S(t) = Psi(0,t) diag(s^2, 0) Psi(0,t)^T + R(t) R(t)^T: the
        second term is the exact eps of Gaussian data x0 ~ N(0, s^2 I) (score = -S^-1 u, eps = -R^T score), whose probability
        flow carries the prior to images of standard deviation s = 0.5 ([0, 1] units: 0.5 +- 0.25) along a bounded trajectory.
        It enters through the reference's own `mixed_score` hook (cld_jax/models/utils.py:174-176: eps_fn adds a matrix times
        u), here with X(t) instead of R(t)^-1 diag(0, 1).  The network (random weights, head convolution scaled by GAIN)
        perturbs that trajectory; GAIN is chosen so that the network moves the final image by far more than the bar
        (asserted) while |u| stays O(1..10).
  blur  eps_y(y, t) = DCT(NCSNpp(IDCT y, t)) + b(t) (.) y, b = sigma / (m^2 s^2 + sigma^2) per frequency: the same construction
        (y0 ~ N(0, s^2 I)) through the reference's `sde.model2eps` hook (blur_jax/sde_lib.py:150-152), exposed by the product
        in linear form.

The oracle side is the float64 restatement loop on a few DISTINCT noises (images are independent units); the batch is built from
those noises laid out over all positions by a seeded random assignment (`_layout`), and every copy of a noise must equal every other
copy BIT FOR BIT.  Each position of the batch is thereby chained to an oracle-checked image: directly where all distinct noises go
through the oracle (config 2), through complementary layouts of further GPU runs where the oracle is too slow for that (config 5).
A defect confined to some batch positions, tile indices or rounds of a persistent walk shows either as a broken copy or -- if it
mixes images up -- as a changed result wherever the two images involved differ (7 of 8 pairs at 8 distinct noises)."""
import copy
import os

import numpy as np
import pytest
import torch

from conftest import oracle_cld_like
from oracle import blur as oblur
from oracle import cld as ocld
from oracle import ncsnpp as onet

pytestmark = pytest.mark.gpu

ABS_TOL = 1e-3  # BASELINE.json north_star: "within fp32 1e-3 per pixel"
S_DATA = 0.5    # standard deviation of the Gaussian data model of the near-denoiser


def _gauss_eps_matrix(sde, t):
    """X(t) = R^T (Psi(0,t) diag(s^2, 0) Psi(0,t)^T + R R^T)^-1 for either the product's or the oracle's CLD object."""
    R = np.asarray(sde.s_R(np.float64(t)), dtype=np.float64)
    P = np.asarray(sde.s_psi(0.0, np.float64(t)), dtype=np.float64)
    S = P @ np.diag([S_DATA * S_DATA, 0.0]) @ P.T + R @ R.T
    return R.T @ np.linalg.inv(S)


def _scaled_head(params, gain):
    p = copy.deepcopy(params)
    p["Conv_1"]["kernel"] = (p["Conv_1"]["kernel"] * gain).astype(np.float32)
    p["Conv_1"]["bias"] = (p["Conv_1"]["bias"] * gain).astype(np.float32)
    return p


def _layout(B, distinct, seed, shift=0):
    """Seeded random assignment of `distinct` noises to the B positions of a batch, every noise used B // distinct times
    (+-1); `shift` rotates the noise ids so that `distinct` shifted layouts put every noise at every position once."""
    lay = np.arange(B) % distinct
    np.random.default_rng(1000 + seed).shuffle(lay)
    return (lay + shift) % distinct


def _assert_copies_identical(tag, lay, *arrays):
    """Every copy of a noise gives the same bits wherever it sits in the batch; returns the position of each noise's first copy."""
    first = {}
    for p, j in enumerate(lay):
        first.setdefault(int(j), p)
    for a in arrays:
        bad = [p for p, j in enumerate(lay) if not np.array_equal(a[p], a[first[int(j)]])]
        assert not bad, "%s: %d of %d positions differ from the first copy of their noise, e.g. position %d (noise %d, first copy at %d): max|d| %.3e" % (
            tag, len(bad), len(lay), bad[0], lay[bad[0]], first[int(lay[bad[0]])], float(np.abs(a[bad[0]] - a[first[int(lay[bad[0]])]]).max()))
    return [first[j] for j in sorted(first)]


def _cld_case(native_lib, cfg_mod, image_size, B, distinct, nfe, gain, seed, method="deis", x64=False, **samp):
    """Returns (cfg, osde, eps64, lin64, u_distinct (float64), run, model): `run(layout)` samples the batch whose position p holds
    noise `layout[p]` and returns (x, v, nfe) as numpy."""
    from gddim_amd.cld import sampling, sde_lib
    from gddim_amd.datasets import get_data_inverse_scaler
    from gddim_amd.models import utils as mutils
    cfg = cfg_mod.get_config()
    cfg.data.image_size = image_size
    cfg.model.x64 = x64  # False = the reference's default: float32 tables on both sides (conftest.oracle_cld_like)
    params = _scaled_head(onet.random_params(cfg, 6, seed=seed), gain)

    class NearDenoiserCLD(sde_lib.CLD):
        def mixed_u_matrix(self, t):  # X(t) instead of the reference's R(t)^-1 diag(0, 1)
            return _gauss_eps_matrix(self, t)

    sde = sde_lib.from_config(cfg)
    sde.__class__ = NearDenoiserCLD
    sde.mixed_score = True
    osde = oracle_cld_like(sde)
    model = mutils.get_model(cfg.model.name)(config=cfg)
    state = mutils.State(params_ema=params, model_state={})
    rng = np.random.default_rng(seed + 100)
    shape = (distinct, image_size, image_size, 3)
    ud = np.stack([rng.standard_normal(shape), rng.standard_normal(shape) / 2], axis=-1).astype(np.float32)
    cfg.sampling.method, cfg.sampling.nfe, cfg.sampling.deis_order, cfg.sampling.noise_removal = method, nfe, 2, True
    for k, v in samp.items():
        setattr(cfg.sampling, k, v)
    fn = sampling.get_sampling_fn(cfg, sde, model, None, get_data_inverse_scaler(cfg))  # the factory: is_p=True, hipGraph replay
    ud_dev = torch.as_tensor(ud, device="cuda:0")

    def run(layout, fill=None):
        """fill: optional (B, H, W, C, 2) noises for the positions whose layout entry is negative (all-distinct batches)."""
        lay = np.asarray(layout)
        u = ud_dev[torch.as_tensor(np.maximum(lay, 0), device="cuda:0", dtype=torch.long)]
        if fill is not None:
            mask = torch.as_tensor(lay < 0, device="cuda:0")
            u = torch.where(mask[:, None, None, None, None], torch.as_tensor(fill, device="cuda:0"), u)
        xs, vs, n = fn([0], state, len(layout), u[None])
        return xs[0].cpu().numpy(), vs[0].cpu().numpy(), n

    net64 = onet.cld_eps_fn(params, cfg, dtype=torch.float64)

    def lin64(uu, t):  # the Gaussian-data model alone: how much of the result is the network's doing
        return ocld.bmm(np.stack([_gauss_eps_matrix(osde, tb) for tb in t]), uu)

    def eps64(uu, t):  # NCSNpp(u, t) + X(t) u
        return net64(uu, t) + lin64(uu, t)

    return cfg, osde, eps64, lin64, ud.astype(np.float64), run, model


def _report(tag, x, v, wx, wv, lx, trace=None):
    ex = float(np.abs(x - wx).max())
    ev = float(np.abs(v - wv).max())
    moved = float(np.abs(wx - lx).max())
    print("%s: max|dx| %.3e (abs, bar %.0e)  max|dv| %.3e  | x in [%.3f, %.3f] std %.3f, |v| <= %.2f, network moved x by %.3f%s"
          % (tag, ex, ABS_TOL, ev, wx.min(), wx.max(), wx.std(), np.abs(wv).max(), moved,
             "" if trace is None else ", max|u| along the trajectory %.2f" % max(float(np.abs(a).max()) for a in trace)))
    return ex, ev, moved


def _oracle(osde, eps64, lin64, u, nfe, trace=None):
    """The float64 restatement loop (cld_jax/sampling.py:204-230) on noises u, with and without the network."""
    wx, wv, _ = ocld.deis_sampler(osde, eps64, u, nfe, 2, ts_order=2, denoising=True, inverse_scaler=ocld.inverse_scaler_centered, trace=trace)
    lx, _, _ = ocld.deis_sampler(osde, lin64, u, nfe, 2, ts_order=2, denoising=True, inverse_scaler=ocld.inverse_scaler_centered)
    return wx, wv, lx


def _chained_runs(tag, run, B, distinct, n_ref, seed):
    """GPU runs over complementary layouts so that EVERY position of the batch holds an oracle-checked noise (ids < n_ref) in one of
    them: run r uses the base layout with the noise ids rotated by r * n_ref.  In every run all copies of a noise must agree bit for
    bit, and a noise must give the same bits in every run (other neighbours, other positions).  Returns the results of the distinct
    noises (D, ...) and the number of runs."""
    runs = -(-distinct // n_ref)
    xd = vd = None
    covered = np.zeros(B, dtype=bool)
    nfe = None
    for r in range(runs):
        lay = _layout(B, distinct, seed, shift=r * n_ref)
        x, v, nfe = run(lay)
        assert np.isfinite(x).all() and np.isfinite(v).all()
        first = _assert_copies_identical("%s run %d" % (tag, r), lay, x, v)
        if xd is None:
            xd, vd = x[first], v[first]
        else:
            assert np.array_equal(x[first], xd) and np.array_equal(v[first], vd), "%s: run %d gives other bits for the same noises" % (tag, r)
        covered |= lay < n_ref
    assert covered.all()
    return xd, vd, nfe, runs


@pytest.mark.parametrize("x64", [False, True], ids=["x64=False", "x64=True"])
def test_cld_nfe50_b256_absolute_parity(native_lib, x64):
    """BASELINE config 2: CLD gDDIM deis_order=2, NFE=50 (49 AB steps + denoising), 256 images, full accr U-Net, hipGraph -- with the
    coefficient tables in the reference's default arithmetic (config.model.x64=False: float32) and in float64, the oracle's tables in
    the same arithmetic each time.  The batch is 8 distinct noises over the 256 positions (seeded random assignment, 32 copies each):
    x64=False sends ALL 8 through the float64 oracle, so every one of the 256 positions is bit-identical to an oracle-checked image;
    x64=True (same kernels, other tables) checks 2 of them and covers the positions with 4 complementary layouts."""
    from gddim_amd.configs import cld_accr_dcifar10
    B, distinct, nfe, gain = 256, 8, 50, 0.3
    n_ref = 2 if x64 else 8
    cfg, osde, eps64, lin64, ud, run, model = _cld_case(native_lib, cld_accr_dcifar10, 32, B, distinct, nfe, gain, seed=0, x64=x64)
    assert osde.dtype == (np.float64 if x64 else np.float32)
    xd, vd, n, runs = _chained_runs("CLD NFE=50 B=256 x64=%s" % x64, run, B, distinct, n_ref, seed=0)
    trace = []
    wx, wv, lx = _oracle(osde, eps64, lin64, ud[:n_ref], nfe, trace)
    ex, ev, moved = _report("CLD NFE=50 B=256 x64=%s (%d noises vs the oracle, %d layouts, 256/256 positions chained)" % (x64, n_ref, runs),
                            xd[:n_ref], vd[:n_ref], wx, wv, lx, trace)
    # the kernels of the benchmark ran (hybrid Winograd kernel, persistent short-K GEMM), not a small-batch fallback
    names = _profile_names(model, 6, 256)
    assert any("conv3x3_wino42_f32_kernel" in k for k in names) and any("igemm_f32_pers_kernel" in k for k in names), names
    assert n == nfe
    assert moved > 20 * ABS_TOL, "the network must matter for the test to mean anything (moved x by %.3e)" % moved
    assert max(float(np.abs(a).max()) for a in trace) < 100.0
    assert ex <= ABS_TOL and ev <= ABS_TOL * max(1.0, float(np.abs(wv).max()))
    if not x64:
        # The copies above cannot see a defect that swaps data between two positions holding the SAME noise (1 pair in 8).
        # So: a batch of 256 DISTINCT noises -- the 8 oracle-checked ones at seeded random positions, 248 fresh ones everywhere else --
        # must reproduce each of the 8 images bit for bit where it sits (any exchange with a neighbour now changes the result), the 8
        # must differ from one another, and no two of the 256 outputs may coincide.
        rng = np.random.default_rng(77)
        pos = rng.choice(B, size=distinct, replace=False)
        lay = np.full(B, -1)
        lay[pos] = np.arange(distinct)
        shape = (B, 32, 32, 3)
        fill = np.stack([rng.standard_normal(shape), rng.standard_normal(shape) / 2], axis=-1).astype(np.float32)
        xa, va, _ = run(lay, fill)
        for j, p_ in enumerate(pos):
            assert np.array_equal(xa[p_], xd[j]) and np.array_equal(va[p_], vd[j]), \
                "noise %d at position %d of an all-distinct batch gives other bits than among copies: max|d| %.3e" % (j, p_, float(np.abs(xa[p_] - xd[j]).max()))
        assert all(not np.array_equal(xd[i], xd[j]) for i in range(distinct) for j in range(i))
        flat = np.ascontiguousarray(xa.reshape(B, -1))
        assert len({row.tobytes() for row in flat}) == B, "two positions of an all-distinct batch returned the same image"
        print("all-distinct batch (8 oracle-checked noises among 248 fresh ones): the 8 images are bit-identical to their copies' result; 256 distinct outputs")


def _profile_names(model, in_ch, B, entries=False):
    """Kernel names one U-Net evaluation at this batch size launches (engine profile of a single forward)."""
    ent = model._engines[(in_ch, 0, 0)]  # (input channels, device, slot)
    eng = ent['eng']
    eng.profile(True)
    xin = torch.zeros((B, eng.image_size, eng.image_size, in_ch), device="cuda:0")
    eng.unet(xin, torch.full((B,), 500.0, device="cuda:0"))
    ents = eng.profile_read()
    eng.profile(False)
    return ents if entries else [p["name"] for p in ents]


def test_cld_nfe100_b256_absolute_parity(native_lib):
    """NFE = 100 (BASELINE config 5's longest schedule) on the 32x32 production batch: error growth over 99 AB steps; 4 distinct
    noises, 2 through the oracle, two complementary layouts."""
    from gddim_amd.configs import cld_accr_dcifar10
    B, distinct, n_ref, nfe, gain = 256, 4, 2, 100, 0.3
    cfg, osde, eps64, lin64, ud, run, model = _cld_case(native_lib, cld_accr_dcifar10, 32, B, distinct, nfe, gain, seed=1)
    xd, vd, n, runs = _chained_runs("CLD NFE=100 B=256", run, B, distinct, n_ref, seed=1)
    wx, wv, lx = _oracle(osde, eps64, lin64, ud[:n_ref], nfe)
    ex, ev, moved = _report("CLD NFE=100 B=256 (%d noises vs the oracle, %d layouts)" % (n_ref, runs), xd[:n_ref], vd[:n_ref], wx, wv, lx)
    assert n == nfe and moved > 20 * ABS_TOL
    assert ex <= ABS_TOL and ev <= ABS_TOL * max(1.0, float(np.abs(wv).max()))


def test_cld_64x64_nfe10_persistent_tile_absolute_parity(native_lib):
    """BASELINE config 5 shape (64x64x3) at a batch that selects the production kernels on 64-wide images (the hybrid Winograd
    kernel on the 64x64 / 32x32 / 16x16 levels: 32 images x 512 tiles / 32 = 512 workgroup tiles x 2 on the top level), NFE = 10.
    x (the image, [0, 1] units) is the quantity under the absolute 1e-3 bar; the velocity v is not an output of the reference's
    pipeline and is bounded by the bar scaled by max|v| (|v| <= 20 on this trajectory)."""
    from gddim_amd.configs import cld_accr_dcifar10
    B, distinct, n_ref, nfe, gain = 32, 2, 1, 10, 0.3
    cfg, osde, eps64, lin64, ud, run, model = _cld_case(native_lib, cld_accr_dcifar10, 64, B, distinct, nfe, gain, seed=2)
    xd, vd, n, runs = _chained_runs("CLD 64x64 NFE=10 B=32", run, B, distinct, n_ref, seed=2)
    wx, wv, lx = _oracle(osde, eps64, lin64, ud[:n_ref], nfe)
    ex, ev, moved = _report("CLD 64x64 NFE=10 B=32", xd[:n_ref], vd[:n_ref], wx, wv, lx)
    names = _profile_names(model, 6, 32)
    assert any("conv3x3_wino42_f32_kernel" in k for k in names), names
    assert n == nfe and moved > 20 * ABS_TOL
    assert ex <= ABS_TOL and ev <= ABS_TOL * max(1.0, float(np.abs(wv).max()))


def test_cld_64x64_b512_full_size_absolute_parity(native_lib):
    """BASELINE config 5 AT ITS STATED SIZE: 64x64x3, B = 512, full accr U-Net, NFE = 10, hipGraph, through get_sampling_fn
    (cld_jax/sampling.py:204-239 at image_size = 64).  The 64x64-level activations are 512 * 4096 * 128 * 4 B = 1.07 GB per tensor
    -- half of the int32 byte-offset range the kernels' buffer addressing lives in.  2 distinct noises over the 512 positions in two
    complementary layouts, noise 0 through the float64 oracle: every position (the far end included, where an offset overflow would
    show) is bit-identical to the oracle-checked image in one of the two runs; and a B = 32 run (another captured graph, other tile
    choices) of the same two noises agrees to round-off.
    (x is the quantity under the north-star bar; v is bounded by the bar scaled by max|v|, |v| <= 20 here.)"""
    from gddim_amd.configs import cld_accr_dcifar10
    B, distinct, n_ref, nfe, gain = 512, 2, 1, 10, 0.3
    cfg, osde, eps64, lin64, ud, run, model = _cld_case(native_lib, cld_accr_dcifar10, 64, B, distinct, nfe, gain, seed=2)
    xd, vd, n, runs = _chained_runs("CLD 64x64 NFE=10 B=512", run, B, distinct, n_ref, seed=2)
    assert n == nfe
    ents = _profile_names(model, 6, B, entries=True)
    names = [p["name"] for p in ents]
    assert any("conv3x3_wino" in k for k in names) and any("igemm_f32_pers_kernel" in k for k in names), names
    # the 256- and 384-channel tensors of the 64x64 level are 2 and 3 GiB at this batch -- beyond the 2 GiB buffer range of the Winograd
    # and fast GEMM kernels: the engine runs those layers as launches over image sub-ranges (engine.cpp: gemm_parts), not on the
    # 64-bit-addressed generic kernel (round 3 found 38 % of this configuration's time there)
    generic = sum(p["flops"] for p in ents if p["name"].startswith("igemm_f32_kernel<"))
    total = sum(p["flops"] for p in ents)
    print("generic-kernel share of the flops at B=512 64x64: %.4f" % (generic / total))
    assert generic < 0.02 * total, [(p["name"], p["flops"]) for p in ents if p["name"].startswith("igemm_f32_kernel<")]
    # the same noises at B = 32 (other kernels per level): round-off apart, not bits
    x32, v32, _ = run(_layout(32, distinct, 7))
    f32 = _assert_copies_identical("CLD 64x64 NFE=10 B=32", _layout(32, distinct, 7), x32, v32)
    far_x, far_v = float(np.abs(x32[f32] - xd).max()), float(np.abs(v32[f32] - vd).max())
    print("CLD 64x64 NFE=10 B=512 vs the same noises at B=32: max|dx| %.3e max|dv| %.3e" % (far_x, far_v))
    assert far_x <= 1e-4 and far_v <= 1e-3, (far_x, far_v)
    wx, wv, lx = _oracle(osde, eps64, lin64, ud[:n_ref], nfe)
    ex, ev, moved = _report("CLD 64x64 NFE=10 B=512 (512/512 positions chained, %d layouts)" % runs, xd[:n_ref], vd[:n_ref], wx, wv, lx)
    assert moved > 20 * ABS_TOL
    assert ex <= ABS_TOL and ev <= ABS_TOL * max(1.0, float(np.abs(wv).max()))
    model.close()  # 35 GB of arena


@pytest.mark.parametrize("nfe", [20, 50, 100])
def test_cld_64x64_b512_nfe_sweep_absolute_parity(native_lib, nfe):
    """BASELINE config 5 at its stated size on its other schedules: 64x64x3, B = 512, full accr U-Net, NFE = 20 / 50 / 100 (other step
    sizes, other DEIS rows, up to ten times the error-growth horizon of the NFE 10 check above), hipGraph, through get_sampling_fn.
    2 distinct noises, noise 0 against the float64 oracle loop; NFE 20 covers all 512 positions with two complementary layouts, NFE 50
    and 100 run one layout (256 positions hold the oracle-checked noise, the other 256 are bit-identical copies of noise 1; the
    position coverage of this batch shape is held by NFE 10 and 20 -- the kernels do not depend on the schedule)."""
    from gddim_amd.configs import cld_accr_dcifar10
    B, distinct, n_ref, gain = 512, 2, 1, 0.3
    cfg, osde, eps64, lin64, ud, run, model = _cld_case(native_lib, cld_accr_dcifar10, 64, B, distinct, nfe, gain, seed=5)
    if nfe <= 20:
        xd, vd, n, runs = _chained_runs("CLD 64x64 NFE=%d B=512" % nfe, run, B, distinct, n_ref, seed=5)
    else:
        lay = _layout(B, distinct, 5)
        x, v, n = run(lay)
        first = _assert_copies_identical("CLD 64x64 NFE=%d B=512" % nfe, lay, x, v)
        xd, vd, runs = x[first], v[first], 1
    assert n == nfe and np.isfinite(xd).all() and np.isfinite(vd).all()
    wx, wv, lx = _oracle(osde, eps64, lin64, ud[:n_ref], nfe)
    ex, ev, moved = _report("CLD 64x64 NFE=%d B=512 (%d layouts)" % (nfe, runs), xd[:n_ref], vd[:n_ref], wx, wv, lx)
    assert moved > 20 * ABS_TOL
    assert ex <= ABS_TOL and ev <= ABS_TOL * max(1.0, float(np.abs(wv).max()))
    model.close()  # 35 GB of arena


def test_hybdeis_absolute_parity(native_lib):
    """`hybdeis` (cld_jax/sampling.py:255-269) through get_sampling_fn against the oracle's restatement of its two-phase grid,
    on a small network, bounded trajectory."""
    from conftest import tiny_config
    from gddim_amd.cld import sampling, sde_lib
    from gddim_amd.datasets import get_data_inverse_scaler
    from gddim_amd.models import utils as mutils
    from oracle import cld_ext as ox
    cfg = tiny_config("cld", nf=16, num_res_blocks=1, image_size=16, ch_mult=(1, 2, 2), attn=(8,))
    params = _scaled_head(onet.random_params(cfg, 6, seed=3), 0.3)
    sde = sde_lib.from_config(cfg)
    osde = oracle_cld_like(sde)
    model = mutils.get_model(cfg.model.name)(config=cfg)
    state = mutils.State(params_ema=params, model_state={})
    rng = np.random.default_rng(4)
    B = 4
    u = np.stack([rng.standard_normal((B, 16, 16, 3)), rng.standard_normal((B, 16, 16, 3)) / 2], axis=-1).astype(np.float32)
    cfg = cfg.copy_and_resolve_references()
    cfg.sampling.method, cfg.sampling.nfe, cfg.sampling.deis_order, cfg.sampling.noise_removal = 'hybdeis', 12, 2, True
    cfg.sampling.noise_nfe_ratio, cfg.sampling.img_t_ratio = 0.3, 0.3
    fn = sampling.get_sampling_fn(cfg, sde, model, None, get_data_inverse_scaler(cfg))
    xs, vs, n = fn([0], state, B, torch.as_tensor(u[None], device="cuda:0"))
    eps64 = onet.cld_eps_fn(params, cfg, dtype=torch.float64)
    wx, wv, wn = ox.hybdeis_sampler(osde, eps64, u.astype(np.float64), 12, 2, noise_nfe_ratio=0.3, img_t_ratio=0.3,
                                    ts_order=cfg.sampling.ts_order, denoising=True, inverse_scaler=ocld.inverse_scaler_centered)
    # plain random-weight network on the non-monotone grid (it restarts at T after the noise phase): the output scale is large,
    # so this comparison is relative to it like the other variant tests
    sx, sv = max(1.0, float(np.abs(wx).max())), max(1.0, float(np.abs(wv).max()))
    ex, ev = float(np.abs(xs[0].cpu().numpy() - wx).max()), float(np.abs(vs[0].cpu().numpy() - wv).max())
    print("hybdeis nfe=12: max|dx| %.3e max|dv| %.3e (|x| %.3e |v| %.3e)" % (ex, ev, sx, sv))
    assert n == wn == 12 and np.isfinite(wx).all() and ex < 1e-4 * sx and ev < 1e-4 * sv


def test_order0_is_em_matches_oracle(native_lib):
    """get_order0_sampler(is_em=True) -> CLD.prepare_naive_coef (cld_jax/sampling.py:171-172, sde_lib.py:276-287)."""
    from conftest import tiny_config
    from gddim_amd.cld import sampling, sde_lib
    from gddim_amd.datasets import get_data_inverse_scaler
    from gddim_amd.models import utils as mutils
    cfg = tiny_config("cld", nf=16, num_res_blocks=1, image_size=16, ch_mult=(1, 2, 2), attn=(8,))
    params = onet.random_params(cfg, 6, seed=5)
    sde = sde_lib.from_config(cfg)
    osde = oracle_cld_like(sde)
    model = mutils.get_model(cfg.model.name)(config=cfg)
    state = mutils.State(params_ema=params, model_state={})
    rng = np.random.default_rng(6)
    B = 2
    u = np.stack([rng.standard_normal((B, 16, 16, 3)), rng.standard_normal((B, 16, 16, 3)) / 2], axis=-1).astype(np.float32)
    cfg = cfg.copy_and_resolve_references()
    cfg.sampling.method, cfg.sampling.nfe, cfg.sampling.is_em, cfg.sampling.noise_removal = 'order0', 8, True, True
    fn = sampling.get_sampling_fn(cfg, sde, model, None, get_data_inverse_scaler(cfg))
    xs, vs, n = fn([0], state, B, torch.as_tensor(u[None], device="cuda:0"))
    eps64 = onet.cld_eps_fn(params, cfg, dtype=torch.float64)
    wx, wv, _ = ocld.order0_sampler(osde, eps64, u.astype(np.float64), 8, denoising=True, inverse_scaler=ocld.inverse_scaler_centered, is_em=True)
    sx, sv = max(1.0, float(np.abs(wx).max())), max(1.0, float(np.abs(wv).max()))
    ex, ev = float(np.abs(xs[0].cpu().numpy() - wx).max()), float(np.abs(vs[0].cpu().numpy() - wv).max())
    print("order0 is_em: max|dx| %.3e max|dv| %.3e (|x| %.3e |v| %.3e)" % (ex, ev, sx, sv))
    assert n == 8 and np.isfinite(wx).all() and ex < 1e-4 * sx and ev < 1e-4 * sv
    # and it is not the exponential-integrator sampler
    cfg.sampling.is_em = False
    x0, _, _ = sampling.get_sampling_fn(cfg, sde, model, None, get_data_inverse_scaler(cfg))([0], state, B, torch.as_tensor(u[None], device="cuda:0"))
    assert (x0 - xs).abs().max().item() > 1e-3 * sx


def test_blur_nfe50_b256_absolute_parity(native_lib):
    """BASELINE config 3: blur gDDIM order0, NFE=50, 256 images, ddpm_deep U-Net on 3 channels, hipGraph; near-denoiser through
    sde.model2eps: eps_y = DCT(NCSNpp(IDCT y, t)) + b(t) (.) y.  Batch layout and position chaining as in the CLD tests."""
    from gddim_amd.blur import sampling, sde_lib
    from gddim_amd.configs import blur_ddpm_deep_cifar10
    from gddim_amd.datasets import get_data_inverse_scaler
    from gddim_amd.models import utils as mutils
    cfg = blur_ddpm_deep_cifar10.get_config()
    gain, B, n_ref, nfe = 0.3, 256, 2, 50
    params = _scaled_head(onet.random_params(cfg, 3, seed=0), gain)

    def b_plane(s_, t):  # sigma / (m^2 s^2 + sigma^2), (32, 32), for the product's or the oracle's SDE object
        tt = np.asarray([t], dtype=np.float64)
        m = np.asarray(s_.y_mean_coef(tt), dtype=np.float64).reshape(32, 32)
        sig = float(np.asarray(s_.y_std_coef(tt)).reshape(-1)[0])
        return sig / (m * m * S_DATA * S_DATA + sig * sig)

    class NearDenoiserSDE(sde_lib.SDE):
        def model2eps_affine(self, t):
            return 1.0, b_plane(self, t)

    sde = sde_lib.from_config(cfg)
    sde.__class__ = NearDenoiserSDE
    osde = oblur.SDE(sigma_blur_max=cfg.model.sigma_blur_max, sampling_eps=cfg.sampling.t0)
    model = mutils.get_model(cfg.model.name)(config=cfg)
    state = mutils.State(params_ema=params, model_state={})
    distinct = 8
    yd = np.random.default_rng(7).standard_normal((distinct, 32, 32, 3)).astype(np.float32)
    cfg.sampling.nfe = nfe
    fn = sampling.get_sampling_fn(cfg, sde, model, None, get_data_inverse_scaler(cfg))
    yd_dev = torch.as_tensor(yd, device="cuda:0")

    def run(layout):
        yy = yd_dev[torch.as_tensor(np.asarray(layout), device="cuda:0", dtype=torch.long)]
        xs, n_ = fn([0], state, len(layout), yy[None])
        a = xs[0].cpu().numpy()
        return a, a, n_

    # 8 distinct noises over the 256 positions, 2 through the oracle, 4 complementary layouts: every position chained (module docstring)
    xd, _, n, runs = _chained_runs("blur NFE=50 B=256", run, B, distinct, n_ref, seed=7)
    x, y = xd, yd
    net64 = onet.blur_eps_fn(params, cfg, dtype=torch.float64)

    def lin64(xx, t):  # IDCT(b (.) DCT x): t is batch-uniform in the sampler
        return osde.y2x(b_plane(osde, float(t[0]))[None, :, :, None] * osde.x2y(xx))

    def eps64(xx, t):
        return net64(xx, t) + lin64(xx, t)

    inv = lambda a: (a + 1) / 2
    trace = []
    wx, wn = oblur.order0_sampler(osde, eps64, y[:n_ref].astype(np.float64), nfe, ts_order=cfg.sampling.ts_order, inverse_scaler=inv, trace=trace)
    lx, _ = oblur.order0_sampler(osde, lin64, y[:n_ref].astype(np.float64), nfe, ts_order=cfg.sampling.ts_order, inverse_scaler=inv)
    err = float(np.abs(x[:n_ref] - wx).max())
    moved = float(np.abs(wx - lx).max())
    print("blur NFE=50 B=256 (2 noises vs the oracle, 4 layouts, 256/256 positions chained): max|dx| %.3e (abs, bar %.0e) | x in [%.3f, %.3f] std %.3f, network moved x by %.3f, max|y| along the trajectory %.2f"
          % (err, ABS_TOL, wx.min(), wx.max(), wx.std(), moved, max(float(np.abs(a).max()) for a in trace)))
    assert n == wn == nfe and np.isfinite(x).all() and moved > 20 * ABS_TOL
    assert err <= ABS_TOL
