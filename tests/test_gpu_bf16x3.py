"""GPU parity of the opt-in split-precision mode (gd_engine_set_precision(e, 1), "bf16x3"): the weight contractions run as
a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on the bf16 matrix core with fp32 accumulation.  Same bars as the fp32 mode."""
import numpy as np
import pytest
import torch

from conftest import oracle_cld_like, tiny_config
from oracle import cld as ocld
from oracle import ncsnpp as onet

pytestmark = pytest.mark.gpu


def test_bf16x3_full_architecture_forward(native_lib):
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.engine import Engine
    cfg = cld_accr_dcifar10.get_config()
    params = onet.random_params(cfg, 6, seed=0)
    rng = np.random.default_rng(1)
    B = 3
    x = rng.standard_normal((B, 32, 32, 6)).astype(np.float32)
    t = rng.uniform(1e-3, 1.0, size=B).astype(np.float32)
    labels = (t * 999).astype(np.float32)
    eng = Engine(cfg, 6, max_batch=B, device=0).load_params(params)
    xt, lt = torch.as_tensor(x, device="cuda:0"), torch.as_tensor(labels, device="cuda:0")
    got32 = eng.unet(xt, lt).cpu().numpy()
    eng.set_precision("bf16x3")
    got = eng.unet(xt, lt).cpu().numpy()
    eng.set_precision("fp32")
    again32 = eng.unet(xt, lt).cpu().numpy()
    eng.close()
    with torch.no_grad():
        want64 = onet.ncsnpp_forward(params, torch.as_tensor(x, dtype=torch.float64), torch.as_tensor(labels, dtype=torch.float64), cfg).numpy()
        want32 = onet.ncsnpp_forward(params, torch.as_tensor(x), torch.as_tensor(labels), cfg).numpy()
    scale = max(1.0, np.abs(want64).max())
    e_x3 = np.abs(got - want64).max() / scale
    e_hip32 = np.abs(got32 - want64).max() / scale
    e_f32 = np.abs(want32 - want64).max() / scale
    print("full arch: |bf16x3-f64| %.3e  |hip fp32-f64| %.3e  |f32 oracle-f64| %.3e" % (e_x3, e_hip32, e_f32))
    assert np.array_equal(got32, again32)            # switching modes back and forth is clean
    assert not np.array_equal(got, got32)            # and the split-precision kernels really ran
    assert e_x3 < max(4 * e_f32, 2e-5), (e_x3, e_f32)


def test_bf16x3_sampler_matches_oracle(native_lib):
    from gddim_amd.cld import sampling, sde_lib
    from gddim_amd.datasets import get_data_inverse_scaler
    from gddim_amd.models import utils as mutils
    cfg = tiny_config("cld", nf=32, num_res_blocks=1, image_size=16, ch_mult=(1, 2, 2), attn=(8,))  # channels % 32 == 0
    params = onet.random_params(cfg, 6, seed=21)
    sde = sde_lib.from_config(cfg)
    osde = oracle_cld_like(sde)
    model = mutils.get_model(cfg.model.name)(config=cfg)
    state = mutils.State(params_ema=params, model_state={})
    rng = np.random.default_rng(0)
    B, nfe = 4, 10
    u = np.stack([rng.standard_normal((B, 16, 16, 3)), rng.standard_normal((B, 16, 16, 3)) / 2], axis=-1).astype(np.float32)
    fn = sampling.get_deis_sampler(sde, model, (16, 16, 3), nfe, get_data_inverse_scaler(cfg), deis_order=2, ts_order=2, denoising=True)
    ut = torch.as_tensor(u, device="cuda:0")
    x32, v32, _ = fn(0, state, B, ut)
    model.engine_for(params, 6, B, 0).set_precision("bf16x3")
    x, v, _ = fn(0, state, B, ut)
    model.engine_for(params, 6, B, 0).set_precision("fp32")
    wx, wv, _ = ocld.deis_sampler(osde, onet.cld_eps_fn(params, cfg, dtype=torch.float64), u.astype(np.float64), nfe, 2, ts_order=2,
                                  denoising=True, inverse_scaler=ocld.inverse_scaler_centered)
    sx, sv = max(1.0, np.abs(wx).max()), max(1.0, np.abs(wv).max())
    ex, ev = np.abs(x.cpu().numpy() - wx).max() / sx, np.abs(v.cpu().numpy() - wv).max() / sv
    e32 = np.abs(x32.cpu().numpy() - wx).max() / sx
    print("bf16x3 sampler: rel err x %.3e v %.3e (fp32 mode %.3e)" % (ex, ev, e32))
    assert not torch.equal(x, x32)
    assert ex < 1e-4 and ev < 1e-4


def test_production_batch_kernels_match_oracle(native_lib):
    """At the benchmark's batch (256) the planner picks other kernels than at the small batches of the tests above: fp32
    128x128 / 128x64 tiles, the split-precision 256x128 kernel on pre-split activations and the halo-staged 3x3 kernel.  Every image
    is an independent unit, so the first three images of a 256-image batch are checked against the float64 oracle of those three
    images (same bars as the small-batch tests), in both precision modes; the profile must show that those kernels really ran."""
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.engine import Engine
    cfg = cld_accr_dcifar10.get_config()
    params = onet.random_params(cfg, 6, seed=0)
    rng = np.random.default_rng(1)
    B, n_ref = 256, 3
    x = rng.standard_normal((B, 32, 32, 6)).astype(np.float32)
    t = rng.uniform(1e-3, 1.0, size=B).astype(np.float32)
    labels = (t * 999).astype(np.float32)
    with torch.no_grad():
        want64 = onet.ncsnpp_forward(params, torch.as_tensor(x[:n_ref], dtype=torch.float64),
                                     torch.as_tensor(labels[:n_ref], dtype=torch.float64), cfg).numpy()
        want32 = onet.ncsnpp_forward(params, torch.as_tensor(x[:n_ref]), torch.as_tensor(labels[:n_ref]), cfg).numpy()
    scale = max(1.0, np.abs(want64).max())
    e_f32 = np.abs(want32 - want64).max() / scale
    eng = Engine(cfg, 6, max_batch=B, device=0).load_params(params)
    xt, lt = torch.as_tensor(x, device="cuda:0"), torch.as_tensor(labels, device="cuda:0")
    seen = {}
    for mode in ("fp32", "bf16x3"):
        eng.set_precision(mode)
        eng.profile(True)
        got = eng.unet(xt, lt).cpu().numpy()
        seen[mode] = {p["name"] for p in eng.profile_read()}
        eng.profile(False)
        err = np.abs(got[:n_ref] - want64).max() / scale
        print("B=256 %s: |hip-f64| %.3e (float32 oracle %.3e)" % (mode, err, e_f32))
        assert np.isfinite(got).all() and err < max(4 * e_f32, 2e-5), (mode, err, e_f32)
        # the same three images alone (small-batch kernels) agree with their slots in the big batch
        small = eng.unet(xt[:n_ref], lt[:n_ref]).cpu().numpy()
        assert np.abs(small - got[:n_ref]).max() / scale < 2e-5
    eng.close()
    # short-K GEMMs (1x1 convolutions, NIN, attention products) on the persistent kernel; the 3x3 stride-2 pyramid convolution that
    # ends on the 4x4 level on the direct split-K tile
    assert "igemm_f32_pers_kernel<128,128,2,4>" in seen["fp32"] and "igemm_f32_fast_kernel<128,64,4,2,false>" in seen["fp32"]
    assert "igemm_f32_fast_kernel<64,64,2,2,false,false,4>" in seen["fp32"]
    # the 3x3 stride-1 convolutions of the 32x32 / 16x16 / 8x8 levels run through the hybrid Winograd kernel F(2,3)y x F(4,3)x (round 3),
    # those of the 4x4 level through the F(2x2,3x3) kernel's channel-split launch + finishing pass; none is left on the direct kernel
    assert "conv3x3_wino42_f32_kernel" in seen["fp32"]
    assert "conv3x3_wino2d_splitk_f32_kernel" in seen["fp32"]
    assert "igemm_bf16x3_kernel<256,128,4,2>" in seen["bf16x3"]
    assert "igemm_bf16x3_kernel<256,128,4,2,act>" in seen["bf16x3"]  # attention products with both operands split in the kernel
    assert "conv3x3_halo_bf16x3_kernel<gn>" in seen["bf16x3"] and "gn_table" in seen["bf16x3"]  # the fused-GroupNorm variant
    assert any(n.startswith("igemm_bf16x3_kernel<128,128,2,4,pre") for n in seen["bf16x3"])


def test_unfused_halo_path_agrees_with_fused(native_lib, tmp_path):
    """GDDIM_X3_NOFUSE=1 (read once per process) keeps the separate GroupNorm-apply pass in front of the halo kernel: the fallback
    of the fused variant.  Both paths apply the same arithmetic to the same values, so their outputs must agree to roundoff."""
    import os
    import subprocess
    import sys
    script = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from gddim_amd.configs import cld_accr_dcifar10
from gddim_amd.engine import Engine
from oracle import ncsnpp as onet
cfg = cld_accr_dcifar10.get_config()
params = onet.random_params(cfg, 6, seed=0)
rng = np.random.default_rng(1)
B = 256
x = rng.standard_normal((B, 32, 32, 6)).astype(np.float32)
labels = (rng.uniform(1e-3, 1.0, size=B) * 999).astype(np.float32)
eng = Engine(cfg, 6, max_batch=B, device=0).load_params(params)
eng.set_precision("bf16x3")
eng.profile(True)
out = eng.unet(torch.as_tensor(x, device="cuda:0"), torch.as_tensor(labels, device="cuda:0")).cpu().numpy()
names = sorted(p["name"] for p in eng.profile_read())
np.savez(sys.argv[1], out=out[:8], names=np.array(names))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for tag, env in (("fused", {}), ("unfused", {"GDDIM_X3_NOFUSE": "1"})):
        f = str(tmp_path / (tag + ".npz"))
        subprocess.run([sys.executable, "-c", script, f], check=True, env=dict(os.environ, **env))
        outs[tag] = np.load(f)
    fused_names, unfused_names = set(outs["fused"]["names"]), set(outs["unfused"]["names"])
    assert "conv3x3_halo_bf16x3_kernel<gn>" in fused_names and "conv3x3_halo_bf16x3_kernel<gn>" not in unfused_names
    assert "conv3x3_halo_bf16x3_kernel" in unfused_names
    a, b = outs["fused"]["out"], outs["unfused"]["out"]
    assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("B", [17, 48, 100])
def test_kernel_selection_boundaries(native_lib, B):
    """The planner's tile choice depends on the batch (workgroup-count thresholds, ragged M tiles at the 4x4 level, the halo kernel
    needing >= 256 workgroups, ...).  Images are independent units: the first two images of a batch of B must agree with the same two
    images evaluated alone (kernel set checked against the oracle above), in both precision modes."""
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.engine import Engine
    cfg = cld_accr_dcifar10.get_config()
    params = onet.random_params(cfg, 6, seed=0)
    rng = np.random.default_rng(B)
    x = torch.as_tensor(rng.standard_normal((B, 32, 32, 6)).astype(np.float32), device="cuda:0")
    labels = torch.as_tensor((rng.uniform(1e-3, 1.0, size=B) * 999).astype(np.float32), device="cuda:0")
    eng = Engine(cfg, 6, max_batch=B, device=0).load_params(params)
    for mode in ("fp32", "bf16x3"):
        eng.set_precision(mode)
        big = eng.unet(x, labels)
        small = eng.unet(x[:2], labels[:2])
        scale = max(1.0, small.abs().max().item())
        assert torch.isfinite(big).all()
        assert (big[:2] - small).abs().max().item() < 3e-5 * scale, (mode, B)
        one = eng.unet(x[B - 1:], labels[B - 1:])  # the last image too (tail tiles)
        assert (big[B - 1:] - one).abs().max().item() < 3e-5 * scale, (mode, B)
    eng.close()
