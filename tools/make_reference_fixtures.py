#!/usr/bin/env python
"""Pin the oracle (and through it the HIP path) against the REFERENCE ITSELF: run qsh-zh/gDDIM and write its outputs as fixtures.

This script cannot run in the build container or on the GPU box (jax 0.2.8 / jaxlib 0.1.59 / flax 0.3.1 / ml-collections / jammy /
tensorflow are absent there, no network).  It is meant for ONE run inside the reference's own image (`/root/reference/Dockerfile:1-15`,
CPU is enough: `JAX_PLATFORM_NAME=cpu`), with the reference checked out as a git repository (its `git_rootdir()` cache paths need one):

    cd <this repository>
    python tools/make_reference_fixtures.py --reference /path/to/gDDIM --out tests/golden

It imports `cld_jax` and `blur_jax` from the reference, calls their public functions -- nothing of the reference's source is copied
or stored -- and writes ONLY DATA (inputs, seeds, outputs, the parameter tree through the reference's own `save_checkpoint`):

    ref_cld_tables.npz      rev_ts and CLD.get_deis_coef(order 2) for NFE 10 / 20 / 50 / 100 (cld_jax/sde_lib.py:308-319, sampling.py:241-249),
                            the float32 R(t) at the SURVEY B.4 anchors (sde_lib.py:93-118), Psi, F, G, invR samples
    ref_tiny_cld.npz        a tiny NCSN++ (accr switches, nf=16, 1 res block, 32x32 -> 16x16, attention at 16): inputs (u, t), eps_fn output
                            (models/utils.py:168-182), a 5-step get_deis_sampler run with INJECTED u (sampling.py:204-253): x, v, nfe
    ref_tiny_cld_ckpt/checkpoint_1   the flax State of that network, written by flax.training.checkpoints.save_checkpoint
    ref_tiny_blur.npz       the same for blur_jax: yeps_fn output and one `order0` run with injected y (blur_jax/sampling.py:53-89)
    ref_tiny_blur_ckpt/checkpoint_1

tests/test_reference_fixtures.py loads these files -- the checkpoints through gddim_amd.checkpoints.restore_checkpoint, i.e. without
flax -- and compares the oracle (CPU) and the HIP path (GPU) with them; it skips while they are absent.  Once they exist, SURVEY App. D's
[memory] assumptions (flax GroupNorm eps, auto-names, msgpack layout, channel packing) are checked facts and `parity` is pinned.
"""
import argparse
import importlib
import os
import sys

import numpy as np


def _import_from(pkg_dir):
    """Import the reference's flat modules (`import sampling`, `import sde_lib`, ... as its own code does) from one package dir."""
    for name in list(sys.modules):
        if name.split(".")[0] in ("sampling", "sde_lib", "utils", "datasets", "losses", "models", "configs", "deis", "blur", "fft", "multistep"):
            del sys.modules[name]
    sys.path.insert(0, pkg_dir)
    mods = {n: importlib.import_module(n) for n in ("sampling", "sde_lib", "utils", "datasets", "losses")}
    mods["mutils"] = importlib.import_module("models.utils")
    importlib.import_module("models.ncsnpp")  # registers 'ncsnpp'
    sys.path.pop(0)
    return mods


def _tiny(config):
    """The accr / ddpm_deep switches on a network small enough for a fixture (the fields tests/conftest.py: tiny_config mirrors)."""
    m = config.model
    m.nf = 16
    m.ch_mult = (1, 2)
    m.num_res_blocks = 1
    m.attn_resolutions = (16,)
    m.init_scale = 1.0   # flax default init_scale = 0 makes eps ~ 1e-10: a vacuous vector (SURVEY D-9)
    config.data.image_size = 32
    if hasattr(m, "used_cache"):
        m.used_cache = False
    return config


def _state(mods, config, seed):
    import jax
    rng = jax.random.PRNGKey(seed)
    rng, step_rng = jax.random.split(rng)
    model, model_state, params = mods["mutils"].init_model(step_rng, config)
    # biases / GroupNorm offsets are zeros after init: perturb every leaf so that a swapped or dropped parameter shows
    leaves, treedef = jax.tree_util.tree_flatten(params)
    nrng = np.random.default_rng(seed + 1)
    leaves = [np.asarray(l) + 0.05 * nrng.standard_normal(np.shape(l)).astype(np.float32) for l in leaves]
    params = jax.tree_util.tree_unflatten(treedef, [jax.numpy.asarray(l) for l in leaves])
    optimizer = mods["losses"].get_optimizer(config).create(params)
    state = mods["mutils"].State(step=1, optimizer=optimizer, lr=config.optim.lr, model_state=model_state,
                                 ema_rate=config.model.ema_rate, params_ema=params, rng=rng)
    return model, state


def _config_record(config):
    m = config.model
    return dict(nf=int(m.nf), ch_mult=np.asarray(m.ch_mult), num_res_blocks=int(m.num_res_blocks),
                attn_resolutions=np.asarray(m.attn_resolutions), image_size=int(config.data.image_size),
                fir=bool(m.fir), skip_rescale=bool(m.skip_rescale), resblock_type=str(m.resblock_type),
                progressive_input=str(m.progressive_input), embedding_type=str(m.embedding_type), fourier_scale=float(m.fourier_scale))


def make_cld(ref, out):
    import jax
    import jax.numpy as jnp
    from flax.training import checkpoints
    mods = _import_from(os.path.join(ref, "cld_jax"))
    sys.path.insert(0, os.path.join(ref, "cld_jax"))
    config = importlib.import_module("configs.accr_dcifar10_config").get_config()
    sys.path.pop(0)
    config.model.used_cache = False
    sde = mods["sde_lib"].from_config(config)

    # ---- tables (full-size SDE: R_dt = 1e-6, RK4, float32 as the reference runs it) ----
    tab = {}
    for nfe in (10, 20, 50, 100):
        rev = mods["sampling"].get_rev_ts(sde, 2, nfe - 1)
        tab["rev_ts_nfe%d" % nfe] = np.asarray(rev)
        tab["deis_coef_nfe%d" % nfe] = np.asarray(sde.get_deis_coef(2, rev, used_cache=False))
    at = np.array([1e-3, 0.1, 0.5, 1.0], np.float32)
    tab["R_at"] = at
    tab["R_values"] = np.stack([np.asarray(sde.s_R(jnp.asarray(t))) for t in at])
    tab["invR_values"] = np.stack([np.asarray(sde.s_invR(jnp.asarray(t))) for t in at])
    tab["F_values"] = np.stack([np.asarray(sde.s_F(jnp.asarray(t))) for t in at])
    tab["G_values"] = np.stack([np.asarray(sde.s_G(jnp.asarray(t))) for t in at])
    tab["psi_pairs"] = np.array([[1.0, 0.796382], [0.5, 0.1], [0.019382, 0.001]], np.float32)
    tab["psi_values"] = np.stack([np.asarray(sde.s_psi(jnp.asarray(s), jnp.asarray(t))) for s, t in tab["psi_pairs"]])
    np.savez_compressed(os.path.join(out, "ref_cld_tables.npz"), **tab)

    # ---- tiny network: eps_fn and a 5-step sampler run with injected noise ----
    config = _tiny(config)
    model, state = _state(mods, config, seed=7)
    ckdir = os.path.join(out, "ref_tiny_cld_ckpt")
    os.makedirs(ckdir, exist_ok=True)
    checkpoints.save_checkpoint(ckdir, state, step=1, keep=1, overwrite=True)
    nrng = np.random.default_rng(11)
    B, nfe, order = 2, 5, 2
    u = np.stack([nrng.standard_normal((B, 32, 32, 3)), nrng.standard_normal((B, 32, 32, 3)) / 2], axis=-1).astype(np.float32)
    t = np.array([0.7, 0.05], np.float32)
    eps_fn = mods["mutils"].get_eps_fn(sde, model, state.params_ema, state.model_state, train=False, continuous=True)
    eps = np.asarray(eps_fn(jnp.asarray(u), jnp.asarray(t)))
    data_shape = mods["utils"].get_data_shape(config)
    inverse_scaler = mods["datasets"].get_data_inverse_scaler(config)
    res = {}
    for denoise in (False, True):
        fn = mods["sampling"].get_deis_sampler(sde, model, data_shape, nfe, inverse_scaler, deis_order=order, ts_order=2,
                                                denoising=denoise, is_p=False)
        x, v, n = fn(jax.random.PRNGKey(0), state, B, jnp.asarray(u))
        res["x_denoise%d" % denoise], res["v_denoise%d" % denoise], res["nfe_denoise%d" % denoise] = np.asarray(x), np.asarray(v), int(n)
    np.savez_compressed(os.path.join(out, "ref_tiny_cld.npz"), u=u, t=t, eps=eps, nfe=nfe, deis_order=order, **res, **_config_record(config))


def make_blur(ref, out):
    import jax
    import jax.numpy as jnp
    from flax.training import checkpoints
    mods = _import_from(os.path.join(ref, "blur_jax"))
    sys.path.insert(0, os.path.join(ref, "blur_jax"))
    config = importlib.import_module("configs.ddpm_deep_cifar10_config").get_config()
    sys.path.pop(0)
    config = _tiny(config)
    sde = mods["sde_lib"].from_config(config)
    model, state = _state(mods, config, seed=9)
    ckdir = os.path.join(out, "ref_tiny_blur_ckpt")
    os.makedirs(ckdir, exist_ok=True)
    checkpoints.save_checkpoint(ckdir, state, step=1, keep=1, overwrite=True)
    nrng = np.random.default_rng(13)
    B, nfe = 2, 6
    y = nrng.standard_normal((B, 32, 32, 3)).astype(np.float32)
    t = np.array([0.6, 0.02], np.float32)
    yeps_fn = mods["mutils"].get_yeps_fn(sde, model, state.params_ema, state.model_state, train=False, continuous=True)
    yeps = np.asarray(yeps_fn(jnp.asarray(y), jnp.asarray(t)))
    config.sampling.nfe = nfe
    data_shape = mods["utils"].get_data_shape(config)
    inverse_scaler = mods["datasets"].get_data_inverse_scaler(config)
    fn = mods["sampling"].get_order0_sampler(sde, model, data_shape, config.sampling.ts_order, nfe, inverse_scaler, is_p=False)
    x, n = fn(jax.random.PRNGKey(0), state, B, jnp.asarray(y))
    tt = np.array([1e-5, 0.1, 0.5, float(sde.sampling_T)], np.float32)
    np.savez_compressed(os.path.join(out, "ref_tiny_blur.npz"), y=y, t=t, yeps=yeps, nfe=nfe, ts_order=float(config.sampling.ts_order),
                        x=np.asarray(x), nfe_out=int(n), sde_t=tt,
                        mean_coef=np.stack([np.asarray(sde.y_mean_coef(jnp.asarray([a]))).reshape(32, 32) for a in tt]),
                        std_coef=np.array([float(np.asarray(sde.y_std_coef(jnp.asarray([a]))).reshape(-1)[0]) for a in tt]),
                        sigma_blur_max=float(config.model.sigma_blur_max), t0=float(config.sampling.t0), **_config_record(config))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    ap.add_argument("--only", choices=["cld", "blur"], default=None)
    a = ap.parse_args()
    try:
        import jax  # noqa: F401
        import flax  # noqa: F401
    except ImportError as e:
        sys.exit("make_reference_fixtures: %s -- run this inside the reference's image (reference Dockerfile:1-15); see the docstring" % e)
    os.makedirs(a.out, exist_ok=True)
    if a.only in (None, "cld"):
        make_cld(a.reference, a.out)
    if a.only in (None, "blur"):
        make_blur(a.reference, a.out)
    print("wrote reference fixtures to", a.out)


if __name__ == "__main__":
    main()
