"""Parity against outputs of the REFERENCE ITSELF (qsh-zh/gDDIM run under jax/flax), when its fixtures exist.

tools/make_reference_fixtures.py produces tests/golden/ref_*.npz and ref_tiny_*_ckpt/ inside the reference's own image (jax 0.2.8 /
flax 0.3.1, reference Dockerfile:1-15); neither that image nor a network is available in the build container or on the GPU box, so
until somebody runs the script these tests SKIP and the oracle stays "parity unpinned" (oracle/__init__.py).  Only data
travels: arrays in .npz files and the parameter tree in the flax msgpack checkpoint the reference wrote, read here WITHOUT flax through
gddim_amd.checkpoints.restore_checkpoint.  The CPU tests check the oracle and the product's host mathematics, the GPU tests the HIP path.

Tolerances: the reference computes its tables and its network in float32 (x64=False); the oracle's and the product's tables are run in
that arithmetic too (float32 scan, grids, Riemann sums), the oracle's network in float64.  Two float32 executions of the table code still
differ by the order XLA accumulates the 10 000-term sums in -- ~3e-5 absolute on entries up to 16.6, as much as float32 differs from
float64 (tests/test_oracle_cld.py::test_float32_scan_tracks_float64) --, network outputs agree to ~1e-5 of their scale; bars are set a few times above that and far below any structural difference (a wrong GroupNorm eps,
channel order or tap order shows at 1e-2 .. 1)."""
import os

import numpy as np
import pytest
import torch

from conftest import tiny_config

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _need(*names):
    paths = [os.path.join(GOLD, n) for n in names]
    missing = [p for p in paths if not os.path.exists(p)]
    if missing:
        pytest.skip("reference fixture(s) absent: %s -- produce them with tools/make_reference_fixtures.py inside the reference's image"
                    % ", ".join(os.path.basename(m) for m in missing))
    return paths


def _cfg(d, kind):
    cfg = tiny_config(kind, nf=int(d["nf"]), num_res_blocks=int(d["num_res_blocks"]), image_size=int(d["image_size"]),
                      ch_mult=tuple(int(c) for c in d["ch_mult"]), attn=tuple(int(a) for a in d["attn_resolutions"]))
    assert bool(d["fir"]) and bool(d["skip_rescale"]) and str(d["resblock_type"]) == "biggan" and str(d["embedding_type"]) == "fourier"
    return cfg


def _params(ckdir):
    from gddim_amd import checkpoints as ck
    from gddim_amd.models.utils import State
    st = ck.restore_checkpoint(ckdir, State())
    assert st.params_ema is not None, "the reference checkpoint did not restore"
    return st


# ---- CPU: oracle and host mathematics ---------------------------------------------------------------------------------------------
def test_tables_match_reference():
    """get_rev_ts, CLD.get_deis_coef (order 2, NFE 10 / 20 / 50 / 100), R(t), R^-1, F, G, Psi: oracle and product vs the reference."""
    (path,) = _need("ref_cld_tables.npz")
    d = np.load(path)
    from gddim_amd.cld import sde_lib as psde
    from gddim_amd.configs import cld_accr_dcifar10
    from oracle import cld as ocld
    osde = ocld.CLD(is_R_rk=True, R_dt=1e-6, dtype=np.float32)
    prod = psde.from_config(cld_accr_dcifar10.get_config())
    assert prod.x64 is False
    for nfe in (10, 20, 50, 100):
        rev_ref, coef_ref = d["rev_ts_nfe%d" % nfe], d["deis_coef_nfe%d" % nfe]
        rev = ocld.get_rev_ts(osde, 2, nfe - 1)
        assert np.allclose(rev, rev_ref, rtol=2e-6, atol=1e-7)
        coef = osde.get_deis_coef(2, rev)
        assert coef.shape == coef_ref.shape
        assert np.abs(coef - coef_ref).max() < 5e-4, (nfe, np.abs(coef - coef_ref).max())
        from gddim_amd.cld import sampling as psamp
        prev = np.asarray(psamp.get_rev_ts(prod, 2, nfe - 1), dtype=np.float64)
        pcoef = np.asarray(prod.get_deis_coef(2, prev), dtype=np.float64)
        assert np.abs(pcoef - coef_ref).max() < 5e-4
    for name, fo, fp in (("R_values", osde.s_R, prod.s_R), ("invR_values", osde.s_invR, prod.s_invR),
                         ("F_values", osde.s_F, prod.s_F), ("G_values", osde.s_G, prod.s_G)):
        for t, want in zip(d["R_at"], d[name]):
            assert np.abs(np.asarray(fo(np.float64(t))) - want).max() < 2e-4 * max(1.0, np.abs(want).max()), (name, t)
            assert np.abs(np.asarray(fp(np.float64(t))) - want).max() < 2e-4 * max(1.0, np.abs(want).max()), (name, t)
    for (s, t), want in zip(d["psi_pairs"], d["psi_values"]):
        assert np.abs(np.asarray(osde.s_psi(np.float64(s), np.float64(t))) - want).max() < 1e-4 * max(1.0, np.abs(want).max())


def test_oracle_network_and_sampler_match_reference_cld():
    """The flax checkpoint restores without flax; oracle eps_fn and the 5-step deis sampler (injected u) vs the reference's outputs."""
    npz, ckdir = _need("ref_tiny_cld.npz", "ref_tiny_cld_ckpt")
    d = np.load(npz)
    from oracle import cld as ocld
    from oracle import ncsnpp as onet
    cfg = _cfg(d, "cld")
    st = _params(ckdir)
    eps_fn = onet.cld_eps_fn(st.params_ema, cfg, dtype=torch.float64)
    got = eps_fn(d["u"].astype(np.float64), d["t"].astype(np.float64))
    sc = max(1.0, float(np.abs(d["eps"]).max()))
    assert np.abs(got - d["eps"]).max() < 1e-4 * sc, np.abs(got - d["eps"]).max() / sc
    osde = ocld.CLD(is_R_rk=cfg.model.is_R_rk, R_dt=cfg.model.R_dt, dtype=np.float32)
    for denoise in (0, 1):
        x, v, n = ocld.deis_sampler(osde, eps_fn, d["u"].astype(np.float64), int(d["nfe"]), int(d["deis_order"]), ts_order=2,
                                    denoising=bool(denoise), inverse_scaler=ocld.inverse_scaler_centered)
        wx, wv = d["x_denoise%d" % denoise], d["v_denoise%d" % denoise]
        assert n == int(d["nfe_denoise%d" % denoise])
        assert np.abs(x - wx).max() < 1e-3 * max(1.0, np.abs(wx).max()) and np.abs(v - wv).max() < 1e-3 * max(1.0, np.abs(wv).max())


def test_oracle_network_and_sampler_match_reference_blur():
    npz, ckdir = _need("ref_tiny_blur.npz", "ref_tiny_blur_ckpt")
    d = np.load(npz)
    from oracle import blur as oblur
    from oracle import ncsnpp as onet
    cfg = _cfg(d, "blur")
    st = _params(ckdir)
    osde = oblur.SDE(sigma_blur_max=float(d["sigma_blur_max"]), sampling_eps=float(d["t0"]))
    for a, mean, std in zip(d["sde_t"], d["mean_coef"], d["std_coef"]):
        tt = np.asarray([a], dtype=np.float64)
        assert np.abs(np.asarray(osde.y_mean_coef(tt)).reshape(32, 32) - mean).max() < 2e-5
        assert abs(float(np.asarray(osde.y_std_coef(tt)).reshape(-1)[0]) - float(std)) < 2e-5
    x_eps = onet.blur_eps_fn(st.params_ema, cfg, dtype=torch.float64)
    # yeps_fn = DCT(eps_fn(IDCT y)) (blur_jax/models/utils.py:155-160)
    got = osde.x2y(x_eps(osde.y2x(d["y"].astype(np.float64)), d["t"].astype(np.float64)))
    sc = max(1.0, float(np.abs(d["yeps"]).max()))
    assert np.abs(got - d["yeps"]).max() < 1e-4 * sc
    x, n = oblur.order0_sampler(osde, x_eps, d["y"].astype(np.float64), int(d["nfe"]), ts_order=float(d["ts_order"]),
                                inverse_scaler=lambda a: (a + 1) / 2)
    assert n == int(d["nfe_out"]) and np.abs(x - d["x"]).max() < 1e-3 * max(1.0, np.abs(d["x"]).max())


# ---- GPU: the HIP path through the reference's API ------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_path_matches_reference_cld(native_lib):
    npz, ckdir = _need("ref_tiny_cld.npz", "ref_tiny_cld_ckpt")
    d = np.load(npz)
    from gddim_amd.cld import sampling, sde_lib
    from gddim_amd.datasets import get_data_inverse_scaler
    from gddim_amd.models import utils as mutils
    cfg = _cfg(d, "cld")
    st = _params(ckdir)
    sde = sde_lib.from_config(cfg)
    model = mutils.get_model(cfg.model.name)(config=cfg)
    state = mutils.State(params_ema=st.params_ema, model_state={})
    u = torch.as_tensor(d["u"], device="cuda:0")
    eps = mutils.get_eps_fn(sde, model, state.params_ema, state.model_state, train=False, continuous=True)(u, torch.as_tensor(d["t"], device="cuda:0"))
    sc = max(1.0, float(np.abs(d["eps"]).max()))
    assert np.abs(eps.cpu().numpy() - d["eps"]).max() < 1e-4 * sc
    n_img = int(d["image_size"])
    for denoise in (0, 1):
        fn = sampling.get_deis_sampler(sde, model, (n_img, n_img, 3), int(d["nfe"]), get_data_inverse_scaler(cfg), deis_order=int(d["deis_order"]),
                                       ts_order=2, denoising=bool(denoise), is_p=False)
        x, v, n = fn(0, state, u.shape[0], u)
        wx, wv = d["x_denoise%d" % denoise], d["v_denoise%d" % denoise]
        assert n == int(d["nfe_denoise%d" % denoise])
        assert np.abs(x.cpu().numpy() - wx).max() < 1e-3 * max(1.0, np.abs(wx).max())
        assert np.abs(v.cpu().numpy() - wv).max() < 1e-3 * max(1.0, np.abs(wv).max())


@pytest.mark.gpu
def test_hip_path_matches_reference_blur(native_lib):
    npz, ckdir = _need("ref_tiny_blur.npz", "ref_tiny_blur_ckpt")
    d = np.load(npz)
    from gddim_amd.blur import sampling, sde_lib
    from gddim_amd.datasets import get_data_inverse_scaler
    from gddim_amd.models import utils as mutils
    cfg = _cfg(d, "blur")
    cfg.sampling.nfe = int(d["nfe"])
    st = _params(ckdir)
    sde = sde_lib.from_config(cfg)
    model = mutils.get_model(cfg.model.name)(config=cfg)
    state = mutils.State(params_ema=st.params_ema, model_state={})
    y = torch.as_tensor(d["y"], device="cuda:0")
    yeps = mutils.get_yeps_fn(sde, model, state.params_ema, state.model_state, train=False, continuous=True)(y, torch.as_tensor(d["t"], device="cuda:0"))
    assert np.abs(yeps.cpu().numpy() - d["yeps"]).max() < 1e-4 * max(1.0, float(np.abs(d["yeps"]).max()))
    fn = sampling.get_sampling_fn(cfg, sde, model, None, get_data_inverse_scaler(cfg), is_p=False)
    x, n = fn(0, state, y.shape[0], y)
    assert n == int(d["nfe_out"]) and np.abs(x.cpu().numpy() - d["x"]).max() < 1e-3 * max(1.0, np.abs(d["x"]).max())
