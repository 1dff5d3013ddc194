"""GPU parity of the split6 precision mode (gd_engine_set_precision(e, 2), GDDIM_PRECISION=split6; round 6): fp32 mode with the 3x3
convolutions of the 64x64 / 32x32 / 16x16 levels on conv3x3_wino2d_split6_kernel -- F(2x2,3x3) with the component GEMMs in the exact
three-piece bf16 split, bias / per-image bias row / residual / scale / fused GroupNorm statistics in its epilogue.  Same oracles and the
same bars as the fp32 mode (the mode's error is fp32-class: tests/test_gpu_split6.py)."""
import numpy as np
import pytest
import torch

import test_gpu_parity_abs as T
from oracle import ncsnpp as onet

pytestmark = pytest.mark.gpu


def _names(eng, B, size, in_ch=6):
    eng.profile(True)
    eng.unet(torch.zeros((B, size, size, in_ch), device="cuda:0"), torch.full((B,), 500.0, device="cuda:0"))
    ents = eng.profile_read()
    eng.profile(False)
    return {p["name"]: p["launches"] for p in ents}


@pytest.mark.parametrize("B", [3, 256])
def test_split6_mode_full_architecture_forward(native_lib, B):
    """The whole accr NCSN++ in the three modes of one engine: split6 as close to float64 as fp32 mode is (first three images of the
    batch: images are independent), the split6 kernel really ran on the 32x32 and 16x16 levels (74 of the 113 3x3 stride-1 convolutions, 87 % of their flops),
    the hybrid kernel keeps the 8x8 level, and switching modes back and forth is clean."""
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.engine import Engine
    cfg = cld_accr_dcifar10.get_config()
    params = onet.random_params(cfg, 6, seed=0)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, 32, 32, 6)).astype(np.float32)
    t = rng.uniform(1e-3, 1.0, size=B).astype(np.float32)
    labels = (t * 999).astype(np.float32)
    eng = Engine(cfg, 6, max_batch=B, device=0).load_params(params)
    xt, lt = torch.as_tensor(x, device="cuda:0"), torch.as_tensor(labels, device="cuda:0")
    got32 = eng.unet(xt, lt).cpu().numpy()
    eng.set_precision("split6")
    got = eng.unet(xt, lt).cpu().numpy()
    names = _names(eng, B, 32)
    eng.set_precision("fp32")
    again32 = eng.unet(xt, lt).cpu().numpy()
    names32 = _names(eng, B, 32)
    eng.close()
    n = 3
    with torch.no_grad():
        want64 = onet.ncsnpp_forward(params, torch.as_tensor(x[:n], dtype=torch.float64), torch.as_tensor(labels[:n], dtype=torch.float64), cfg).numpy()
        want32 = onet.ncsnpp_forward(params, torch.as_tensor(x[:n]), torch.as_tensor(labels[:n]), cfg).numpy()
    scale = max(1.0, np.abs(want64).max())
    e6, e32, ef = (np.abs(a[:n] - want64).max() / scale for a in (got, got32, want32))
    print("full arch B=%d: |split6-f64| %.3e  |hip fp32-f64| %.3e  |f32 oracle-f64| %.3e; split6 launches %d, hybrid %d (fp32 mode: %d)"
          % (B, e6, e32, ef, sum(v for k, v in names.items() if "split6" in k), sum(v for k, v in names.items() if "wino42" in k),
             sum(v for k, v in names32.items() if "wino42" in k)))
    assert np.isfinite(got).all() and np.array_equal(got32, again32) and not np.array_equal(got, got32)
    assert e6 < max(4 * ef, 2e-5), (e6, ef)
    assert sum(v for k, v in names.items() if "split6" in k) == 74 and not any("split6" in k for k in names32)
    # the GroupNorm of the 16x16 level in the split6 kernel's epilogue where fp32 mode has it in the hybrid kernel's (production batch sizes)
    assert names.get("conv3x3_wino2d_split6_kernel+gn", 0) >= 9 and names.get("gn_apply", 0) <= names32.get("gn_apply", 0) + 2, (names, names32)
    assert any("wino42" in k for k in names)  # the 8x8 level


def test_split6_mode_cld_nfe50_b256_absolute_parity(native_lib, gd_env):
    """BASELINE config 2 in split6 mode at the north-star's bar: CLD gDDIM order 2, NFE = 50, 256 images, full accr U-Net, whole-call
    hipGraph through get_sampling_fn; 4 distinct noises over the 256 positions, 2 through the float64 oracle, two complementary layouts
    (every position chained), copies bit-identical."""
    from gddim_amd.configs import cld_accr_dcifar10
    gd_env(GDDIM_PRECISION="split6")
    B, distinct, n_ref, nfe = 256, 4, 2, 50
    cfg, osde, eps64, lin64, ud, run, model = T._cld_case(native_lib, cld_accr_dcifar10, 32, B, distinct, nfe, 0.3, seed=3)
    xd, vd, n, runs = T._chained_runs("split6 CLD NFE=50 B=256", run, B, distinct, n_ref, seed=3)
    eng = model._engines[(6, 0, 0)]['eng']
    assert eng.plan()["options"]["GDDIM_PRECISION"] == "split6"
    wx, wv, lx = T._oracle(osde, eps64, lin64, ud[:n_ref], nfe)
    ex, ev, moved = T._report("split6 mode, CLD NFE=50 B=256 (%d noises vs the oracle, %d layouts)" % (n_ref, runs), xd[:n_ref], vd[:n_ref], wx, wv, lx)
    names = T._profile_names(model, 6, 256)
    assert any("conv3x3_wino2d_split6_kernel" in k for k in names), names
    assert n == nfe and moved > 20 * T.ABS_TOL
    assert ex <= T.ABS_TOL and ev <= T.ABS_TOL * max(1.0, float(np.abs(wv).max()))
    model.close()


@pytest.mark.parametrize("B", [32, 512])
def test_split6_mode_64x64_nfe10_absolute_parity(native_lib, gd_env, B):
    """BASELINE config 5 (64x64x3, NFE = 10) in split6 mode, at B = 32 and at its full size B = 512 (the 256-channel tensors of the 64x64
    level are 2 GiB there: the engine runs those launches over image sub-ranges): the 64x64 level is tiled by 16 blocks per image."""
    from gddim_amd.configs import cld_accr_dcifar10
    gd_env(GDDIM_PRECISION="split6")
    distinct, n_ref, nfe = 2, 1, 10
    cfg, osde, eps64, lin64, ud, run, model = T._cld_case(native_lib, cld_accr_dcifar10, 64, B, distinct, nfe, 0.3, seed=2)
    xd, vd, n, runs = T._chained_runs("split6 CLD 64x64 NFE=10 B=%d" % B, run, B, distinct, n_ref, seed=2)
    wx, wv, lx = T._oracle(osde, eps64, lin64, ud[:n_ref], nfe)
    ex, ev, moved = T._report("split6 mode, CLD 64x64 NFE=10 B=%d" % B, xd[:n_ref], vd[:n_ref], wx, wv, lx)
    names = T._profile_names(model, 6, B)
    assert any("conv3x3_wino2d_split6_kernel" in k for k in names), names
    assert n == nfe and moved > 20 * T.ABS_TOL
    assert ex <= T.ABS_TOL and ev <= T.ABS_TOL * max(1.0, float(np.abs(wv).max()))
    model.close()
