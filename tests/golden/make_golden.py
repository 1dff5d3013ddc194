#!/usr/bin/env python
"""Generate the golden fixtures in this directory.

IMPORTANT: these vectors are produced by this repository's CPU ORACLE (oracle/), not by the reference: qsh-zh/gDDIM
cannot be imported in this environment (jax / flax / ml_collections / jammy are absent and there is no network) and
its own tests hold no golden vectors.  They pin the oracle against regressions and give the GPU
tests inputs/outputs that do not depend on the oracle being importable.

  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from conftest import tiny_config  # noqa: E402
from oracle import blur as oblur  # noqa: E402
from oracle import cld as ocld  # noqa: E402
from oracle import cld_ext as ox  # noqa: E402
from oracle import rng as orng  # noqa: E402
from oracle import ncsnpp as onet  # noqa: E402


def main():
    osde = ocld.CLD(is_R_rk=True, R_dt=1e-6, dtype=np.float64)
    # 1. coefficient tables of the benchmarked sampler: deis_order=2, ts_order=2, noise_removal=True
    out = {}
    for nfe in (10, 20, 50, 100):
        rev = ocld.get_rev_ts(osde, 2, nfe - 1)
        out["rev_ts_nfe%d" % nfe] = rev
        out["deis_coef_nfe%d" % nfe] = osde.get_deis_coef(2, rev)
    out["R_at"] = np.array([1e-3, 0.1, 0.5, 1.0])
    out["R_values"] = np.stack([osde.s_R(t) for t in out["R_at"]])
    np.savez_compressed(os.path.join(HERE, "cld_deis_coef.npz"), **out)

    # 2. tiny NCSN++ forward + CLD sampler trace (weights regenerated from the seed by oracle.ncsnpp.random_params)
    cfg = tiny_config("cld")
    params = onet.random_params(cfg, 6, seed=11)
    rng = np.random.default_rng(12)
    x = rng.standard_normal((2, 16, 16, 6)).astype(np.float32)
    t = np.array([0.9, 0.05], np.float32)
    with torch.no_grad():
        eps = onet.ncsnpp_forward(params, torch.as_tensor(x, dtype=torch.float64), torch.as_tensor(t * 999, dtype=torch.float64), cfg).numpy()
    u = np.stack([rng.standard_normal((2, 16, 16, 3)), rng.standard_normal((2, 16, 16, 3)) / 2], axis=-1).astype(np.float32)
    trace = []
    sx, sv, _ = ocld.deis_sampler(osde, onet.cld_eps_fn(params, cfg, dtype=torch.float64), u.astype(np.float64), 5, 2, ts_order=2,
                                  denoising=True, inverse_scaler=ocld.inverse_scaler_centered, trace=trace)
    flat = onet.flatten(params)
    np.savez_compressed(os.path.join(HERE, "tiny_cld.npz"), params_seed=11,
                        params_checksum=np.float64(sum(float(np.abs(v).sum()) for v in flat.values())),
                        x=x, t=t, eps=eps, u=u, nfe=5, deis_order=2, sample_x=sx, sample_v=sv, trace=np.stack(trace))

    # 3. tiny blur sampler
    bcfg = tiny_config("blur", nf=16, num_res_blocks=1, image_size=32, ch_mult=(1, 2, 2), attn=(16,))
    bparams = onet.random_params(bcfg, 3, seed=13)
    bsde = oblur.SDE(sigma_blur_max=bcfg.model.sigma_blur_max, sampling_eps=bcfg.sampling.t0)
    y = np.random.default_rng(14).standard_normal((2, 32, 32, 3)).astype(np.float32)
    bx, _ = oblur.order0_sampler(bsde, onet.blur_eps_fn(bparams, bcfg, dtype=torch.float64), y.astype(np.float64), 4, ts_order=2,
                                 inverse_scaler=lambda a: (a + 1) / 2)
    np.savez_compressed(os.path.join(HERE, "tiny_blur.npz"), params_seed=13, y=y, nfe=4, sample_x=bx,
                        rev_ts=oblur.get_rev_ts(bsde, 2, 4), mean_coef=bsde.y_mean_coef(oblur.get_rev_ts(bsde, 2, 4))[..., 0],
                        std_coef=bsde.y_std_coef(oblur.get_rev_ts(bsde, 2, 4)))
    # 4. sampler variants: coefficient tables (R scan with the reference default R_dt = 1e-5) and the counter-based noise
    esde = ocld.CLD(is_R_rk=True, R_dt=1e-5, dtype=np.float64)
    rev = ocld.get_rev_ts(esde, 2, 4)
    lam = ox.LambdaSDE(esde, 1.0, use_order0=True, dt=1e-5)
    ext = {"rev_ts": rev,
           "ldeis_coef_order1": ox.LSDE(esde).get_deis_coef(1, rev),
           "ldeis_eps_map": np.stack([ox.LSDE(esde).s_L(t).T @ ocld.inv_2x2(esde.s_R(np.float64(t)).T) for t in rev[:-1]]),
           "mldeis_coef_order1": ox.MLCLD(esde).get_deis_coef(1, rev),
           "mldeis_psi1": np.stack([ox.s_psi1(esde, t) for t in rev]),
           "sdeis_coef_order0": lam.get_deis_coef(0, rev), "sdeis_coef_order1": lam.get_deis_coef(1, rev),
           "sscs_ou_03_035_cov": ox.sscs_ou(esde, np.zeros((1, 1, 2)), 0.3, 0.35, np.zeros((1, 1, 2)))[1],
           "noise_seed": np.uint64(20240611), "noise_draw3_first8": orng.normal_pairs(20240611, np.arange(8), 3)}
    # an sscs trajectory of the tiny network with that noise source (what the GPU test reproduces through the C ABI)
    sx, sv, _ = ox.sscs_sampler(esde, onet.cld_eps_fn(params, cfg, dtype=torch.float64), u.astype(np.float64), 4, ts_order=2, denoising=True,
                                inverse_scaler=ocld.inverse_scaler_centered,
                                noise_fn=lambda draw, shape: orng.noise_like(20240611, shape, draw))
    ext["sscs_sample_x"], ext["sscs_sample_v"] = sx, sv
    np.savez_compressed(os.path.join(HERE, "cld_variants.npz"), **ext)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
