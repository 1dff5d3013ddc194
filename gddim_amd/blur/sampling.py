"""Blur gDDIM sampler -- drop-in for blur_jax/sampling.py:11-89 on the native MI355X engine.

  get_sampling_fn(config, sde, model, shape, inverse_scaler, is_p=True)     :11-40 (only 'order0' exists)
  get_rev_ts(sde, ts_order, num_step)                                       :42-51
  get_order0_sampler(sde, model, data_shape, ts_order, nfe, inverse_scaler, is_p=False)   :53-89
  get_deis_sampler(sde, model, data_shape, ts_order, nfe, deis_order, inverse_scaler, is_p=False)   -- not in the reference: the
      multistep sampler blur_jax/multistep.py was started for, with per-frequency coefficients;
      `sampling.method = 'deis'` selects it, everything else raises RuntimeError as in the reference

`sampling_fn(rng, state, batch_size, u=None) -> (x, nfe)` (2-tuple, unlike CLD).  The NFE loop
[IDCT -> U-Net -> DCT -> per-frequency DDIM update] runs entirely in `gd_sample_blur`, by default as ONE hipGraph per call.
is_p=True keeps the reference's pmap contract (:82-87): u (n_dev, B, H, W, C) -> xs (n_dev, B, H, W, C), one engine per local GPU
(gddim_amd.devices)."""
import itertools

import numpy as np

from .. import devices
from ..cld.sampling import _seed_of
from . import utils

# 'loop' = one hipGraph for the whole sampler call (default), True = one graph per step replayed nfe times, False = eager launches;
# bit-identical (tests/test_gpu_sampler.py).  Read at call time.
DEFAULT_GRAPH = 'loop'
_TOKENS = itertools.count(1)


def get_rev_ts(sde, ts_order, num_step):
    return np.power(np.linspace(np.power(sde.sampling_T, 1.0 / ts_order), np.power(sde.sampling_eps, 1.0 / ts_order),
                                num_step + 1), ts_order)


def _scale_mode(inverse_scaler):
    probe = np.array([-1.0, 0.0, 1.0])
    scaled = np.asarray(inverse_scaler(probe), dtype=np.float64)
    fuse_scale = bool(np.allclose(scaled, (probe + 1.0) / 2.0))
    return fuse_scale, (None if fuse_scale or np.allclose(scaled, probe) else inverse_scaler)


def _model2eps_rows(sde, rev_ts, nfe):
    """sde.model2eps (blur_jax/sde_lib.py:150-152) is the identity in the reference's SDE class.  A subclass that predicts
    something else may expose the linear form `model2eps_affine(t) -> (a, b)`: eps_x = a * model_output + b * x for a scalar b, or,
    for a (32, 32) plane b in DCT space, eps_y = a * DCT(model_output) + b (.) y -- which the fused loop applies per step.  Any other
    override of model2eps cannot run inside the fused loop and is refused."""
    aff = getattr(sde, 'model2eps_affine', None)
    if aff is not None:
        rows = np.empty((nfe, 1025), dtype=np.float64)
        for i, t in enumerate(rev_ts[:nfe]):
            a, b = aff(float(t))
            rows[i, 0] = a
            rows[i, 1:] = np.broadcast_to(np.asarray(b, dtype=np.float64), (32, 32)).reshape(-1)
        return rows
    probe = np.array([0.25, -2.0])
    if not np.array_equal(np.asarray(sde.model2eps(probe * 3.0, 0.5, probe)), probe):
        raise NotImplementedError("sde.model2eps is not the identity: expose it as model2eps_affine(t) -> (a, b)")
    return None


def get_deis_sampler(sde, model, data_shape, ts_order, nfe, deis_order, inverse_scaler, is_p=False):
    """Adams-Bashforth exponential integrator of order `deis_order` in DCT space: y' = psi_i y + sum_o c_{i,o} eps_o per frequency
    (multistep.get_ab_eps_coef_freq); deis_order = 0 is the order0 sampler."""
    from . import multistep
    rev_ts = get_rev_ts(sde, ts_order=ts_order, num_step=nfe)
    eps_coef, psi = multistep.get_ab_eps_coef_freq(sde, deis_order, rev_ts, deis_order)
    coef = np.concatenate([psi[:, None], eps_coef], axis=1)  # (nfe, deis_order + 2, 32, 32)
    fuse_scale, post_scale = _scale_mode(inverse_scaler)
    token = ("blur_deis", next(_TOKENS))
    m2e = _model2eps_rows(sde, rev_ts, nfe)

    def install(eng):
        eng.set_schedule_blur_deis(rev_ts[:nfe], coef, token=token)
        eng.set_model2eps_blur(m2e)

    return _make_sampler(sde, model, data_shape, nfe, fuse_scale, post_scale, token, install, is_p)


def _make_sampler(sde, model, data_shape, nfe, fuse_scale, post_scale, token, install, is_p):
    """The closure pair of blur_jax/sampling.py:55-89: `install(engine)` uploads this sampler's schedule."""

    def sampler(rng, state, batch_size, u=None, use_graph=None, slot=0):
        import torch
        if use_graph is None:
            use_graph = DEFAULT_GRAPH
        if u is None:
            u = sde.prior_sampling(_seed_of(rng), (batch_size,) + tuple(data_shape))
        if not hasattr(u, 'is_cuda'):
            u = torch.as_tensor(np.asarray(u), dtype=torch.float32)
        if not u.is_cuda:
            u = u.to(torch.device('cuda', torch.cuda.current_device()))
        eng = model.engine_for(state.params_ema, data_shape[-1], batch_size, u.device.index, slot=slot)
        if eng._schedule_token != token:
            install(eng)
        x = eng.sample_blur(u.to(torch.float32), inverse_scale=fuse_scale, use_graph=use_graph)
        return (post_scale(x) if post_scale is not None else x), nfe

    def psampler(prng, pstate, batch_size, u=None, use_graph=None):
        # jax.pmap(sampler, static_broadcasted_argnums=2)(prng, pstate, batch_size, u) (blur_jax/sampling.py:86)
        return devices.pmap_sampler(sampler, prng, pstate, batch_size, u, sde, tuple(data_shape), 1, use_graph=use_graph)

    return psampler if is_p else sampler


def get_order0_sampler(sde, model, data_shape, ts_order, nfe, inverse_scaler, is_p=False):
    rev_ts = get_rev_ts(sde, ts_order=ts_order, num_step=nfe)
    mean_coef = sde.y_mean_coef(rev_ts)[..., 0]  # (nfe+1, 32, 32): identical for every image (t is batch-uniform)
    std_coef = sde.y_std_coef(rev_ts)
    fuse_scale, post_scale = _scale_mode(inverse_scaler)
    token = ("blur_order0", next(_TOKENS))
    m2e = _model2eps_rows(sde, rev_ts, nfe)

    def install(eng):
        eng.set_schedule_blur(rev_ts[:nfe], mean_coef, std_coef, token=token)
        eng.set_model2eps_blur(m2e)

    return _make_sampler(sde, model, data_shape, nfe, fuse_scale, post_scale, token, install, is_p)


def get_sampling_fn(config, sde, model, shape, inverse_scaler, is_p=True):
    del shape
    sampler_name = config.sampling.method
    data_shape = utils.get_data_shape(config)
    if sampler_name.lower() == 'order0':
        return get_order0_sampler(sde=sde, model=model, data_shape=data_shape, ts_order=config.sampling.ts_order,
                                  nfe=config.sampling.nfe, inverse_scaler=inverse_scaler, is_p=is_p)
    if sampler_name.lower() == 'deis':  # extension, see get_deis_sampler
        return get_deis_sampler(sde=sde, model=model, data_shape=data_shape, ts_order=config.sampling.ts_order,
                                nfe=config.sampling.nfe, deis_order=config.sampling.get('deis_order', 1),
                                inverse_scaler=inverse_scaler, is_p=is_p)
    raise RuntimeError
