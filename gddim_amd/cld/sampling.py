"""CLD gDDIM samplers -- drop-in for cld_jax/sampling.py on the native MI355X engine.

Same factory names, argument names and return tuples as the reference:

  get_sampling_fn(config, sde, model, shape, inverse_scaler)           cld_jax/sampling.py:41-154
  get_deis_sampler(sde, model, data_shape, nfe, inverse_scaler, deis_order, ts_order=2,
                   denoising=False, is_p=False)                        :251-253
  get_order0_sampler(sde, model, data_shape, nfe, inverse_scaler, is_em=False,
                     denoising=False, is_p=False)                      :156-202
  get_hyd_deis_sampler(...)                                            :255-269 (different rev_ts only)
  get_rev_ts / get_denoising_step / get_cld_denoising_step             :241-249, :30-39, :23-27

The returned callable is `sampling_fn(rng, state, batch_size, u=None)`:
  is_p=False: u (B,H,W,C,2) -> (x (B,H,W,C), v (B,H,W,C), nfe)
  is_p=True : the reference pmaps over jax.local_device_count() devices from ONE process (:232-237).  Same contract here:
              prng is a stack of n_dev keys, u (n_dev,B,H,W,C,2) -> (xs (n_dev,B,H,W,C), vs, nfe) with n_dev =
              gddim_amd.devices.local_device_count() native engines, one per local GPU, each call one asynchronous
              hipGraphLaunch per engine issued back to back from the calling thread; xs / vs are ShardedDeviceArrays whose
              shards stay on their GPUs until the caller's np.clip(xs * 255., 0, 255) concatenates them on the host
              (cld_jax/run_lib.py:723).  Under a one-process-per-GPU launcher (WORLD_SIZE > 1, gddim_amd.distributed) n_dev is 1.
`u` is the parity hook: with it the deterministic samplers never touch the RNG.

Inside the callable nothing but the native engine runs: the whole NFE loop (U-Net forward + fused
2x2 exponential-integrator update per step, denoising step, inverse scaling) is `gd_sample_cld`, replayed
from a hipGraph.  get_L_deis_sampler (:533-540, the Cholesky parametrisation of eps) and get_mldeis_sampler (:322-378, the
rotated frame y = psi1(t)^-1 x) run on the same fused loop through per-step 2x2 maps (gd_engine_set_transforms_cld), and so does
sde.mixed_score=True (models/utils.py:174-176).  get_em_sampler (:625-655) and get_sscs_sampler (:563-621) add per-step Gaussian
noise from the engine's counter-based generator (same law as the reference's jax.random draws, not the same stream);
get_ode_sampler (:432-495) integrates on the host with the drift on the GPU; get_sdeis_sampler (:380-427) uses the LambdaSDE tables
(sde_lib.py:334-466).  Every name of the reference's factory is built.
"""
import itertools

import numpy as np

from .. import devices
from .._abi import GddimError
from ..models import utils as mutils
from . import utils

# How a sampler call reaches the GPU unless the caller says otherwise: 'loop' = ONE hipGraph for the whole call (pack, every step, the
# denoising evaluation, finish: BASELINE north-star "the outer NFE loop captured as a hipGraph"), True = one graph per step replayed
# num_step times, False = eager launches.  All three are bit-identical (tests/test_gpu_sampler.py); round 5 measured the two captured
# forms equal in steady state (1 849.9 vs 1 851.9 ms per 256-image call), the whole-call graph costing 50 ms more on a batch size's first
# call and 10.5 MB of device memory (tools/graph_first_call.py).  Read at call time (use_graph=None), so it can be changed on the module.
DEFAULT_GRAPH = 'loop'
_TOKENS = itertools.count(1)


def get_cld_denoising_step(sde, t, denoising_eps):
    cur_F = sde.s_F(t)

    def step_fn(u):
        return u + utils.sbmm(cur_F, u) * (-denoising_eps)

    return step_fn


def get_denoising_step(sde, model, t, denoising_eps):
    """u + F u dt - G G score dt at time t with dt = -denoising_eps (cld_jax/sampling.py:30-39).
    Stand-alone (unfused) version built from the operator API; the fused samplers below do this in-kernel."""
    cur_F, cur_G = sde.s_F(t), sde.s_G(t)

    def step_fn(state, u):
        import torch
        eps_fn = mutils.get_eps_fn(sde, model, state.params_ema, state.model_state, train=False, continuous=True)
        _ones = torch.ones(u.shape[0], device=u.device, dtype=torch.float32)
        dt = denoising_eps * (-1)
        cur_eps = eps_fn(u, _ones * t)
        cur_score = sde.eps2score(cur_eps, _ones * t)
        return u + utils.sbmm(cur_F, u) * dt - utils.sbmm(cur_G @ cur_G, cur_score) * dt

    return step_fn


def get_rev_ts(sde, ts_order, num_step):
    """cld_jax/sampling.py:241-249 in the SDE's dtype (float32 for config.model.x64=False): jnp.power of the two Python-float
    end points, jnp.linspace between them, jnp.power(., ts_order) -- which for a Python int exponent is lax.integer_pow, a chain
    of multiplications (jax/_src/numpy/lax_numpy.py `power`, from memory), and libm pow otherwise."""
    from .sde_lib import linspace
    ft = np.dtype(getattr(sde, '_ft', np.float64)).type
    lin = linspace(np.power(ft(sde.T), ft(1.0 / ts_order)), np.power(ft(sde.sampling_eps), ft(1.0 / ts_order)), num_step + 1, ft=ft)
    if isinstance(ts_order, (int, np.integer)) and ts_order >= 0:
        acc, base, y = None, lin, int(ts_order)  # binary exponentiation, as integer_pow lowers
        while y:
            if y & 1:
                acc = base if acc is None else acc * base
            y >>= 1
            if y:
                base = base * base
        return np.ones_like(lin) if acc is None else acc
    return np.power(lin, ft(ts_order))


def _denoise_mats(sde, t):
    F = sde.s_F(t)
    G = sde.s_G(t)
    RinvT = sde.s_invR(t).T
    return F, G @ G, RinvT


def _unreplicate_rng(prng):
    try:
        return prng[0]
    except (TypeError, IndexError):
        return prng


def _seed_of(rng):
    if rng is None:
        return 0
    if isinstance(rng, (int, np.integer)):
        return int(rng)
    if hasattr(rng, 'integers'):
        return int(rng.integers(0, 2 ** 62))
    a = np.asarray(rng.detach().cpu() if hasattr(rng, 'detach') else rng).astype(np.uint64).ravel()
    return int(np.bitwise_xor.reduce(a * np.uint64(0x9E3779B97F4A7C15) + np.arange(a.size, dtype=np.uint64)) >> np.uint64(2))


def _make_sampler(sde, model, data_shape, nfe, inverse_scaler, rev_ts, coef, denoising, is_p, tag, pre=None, eps_lin=None,
                  init=None, post=None, dn_sde=None, noise=None):
    """Shared closure builder: `coef` is (num_step, n_slots, 2, 2) = [Psi_i, C_i0, ...].
    pre / eps_lin: optional (num_step, 2, 2) maps (network input P_i u; eps <- E_i eps); init / post: 2x2 maps applied to the
    initial noise / to the state after the denoising step; dn_sde: the SDE whose F, G, R the denoising step uses;
    noise: (pre_lin, pre_noise, post_noise) per-step maps of the stochastic samplers (Engine.set_noise_cld), seeded per call
    from `rng`."""
    num_step = coef.shape[0]
    dn_sde = sde if dn_sde is None else dn_sde
    dn = _denoise_mats(dn_sde, dn_sde.sampling_eps) if denoising else (None, None, None)
    # mixed score (models/utils.py:174-176): eps_fn(x, t) = net(x, t) + X(t) x with X = invR(t) diag(0, 1), x = P_i u
    u_lin = dn_u_lin = None
    if getattr(sde, 'mixed_score', False):
        base = dn_sde if hasattr(dn_sde, 'mixed_u_matrix') else sde.sde
        X = np.stack([base.mixed_u_matrix(t) for t in rev_ts[:num_step]]) if num_step else np.zeros((0, 2, 2))
        if pre is not None:
            X = X @ pre
        u_lin = X if eps_lin is None else eps_lin @ X
        dn_u_lin = base.mixed_u_matrix(dn_sde.sampling_eps) if denoising else None
    transforms = any(m is not None for m in (pre, eps_lin, u_lin, dn_u_lin, post))
    # the reference applies `inverse_scaler` after the loop; the engine fuses the centered one, (x+1)/2
    probe = np.array([-1.0, 0.0, 1.0])
    scaled = np.asarray(inverse_scaler(probe), dtype=np.float64)
    if np.allclose(scaled, (probe + 1.0) / 2.0):
        fuse_scale, post_scale = True, None
    elif np.allclose(scaled, probe):
        fuse_scale, post_scale = False, None
    else:
        fuse_scale, post_scale = False, inverse_scaler
    token = (tag, next(_TOKENS))
    ts32 = np.asarray(rev_ts[:num_step], dtype=np.float32)
    coef32 = np.ascontiguousarray(coef, dtype=np.float32)

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
        u = u.to(torch.float32)
        assert tuple(u.shape) == (batch_size,) + tuple(data_shape) + (2,), "u must be (B, *data_shape, 2)"
        eng = model.engine_for(state.params_ema, 2 * data_shape[-1], batch_size, u.device.index, slot=slot)
        if init is not None:
            u = utils.sbmm(init, u)
        if eng._schedule_token != token:
            eng.set_schedule_cld(ts32, coef32, denoising, sde.sampling_eps, dn[0], dn[1], dn[2], token=token)
            if transforms:
                eng.set_transforms_cld(pre, eps_lin, u_lin, dn_u_lin, post)
            if noise is not None:
                eng.set_noise_cld(*noise)
        if noise is not None:
            eng.set_seed(_seed_of(rng))
        x, v = eng.sample_cld(u, inverse_scale=fuse_scale, use_graph=use_graph)
        if post_scale is not None:
            x = post_scale(x)
        return x, v, nfe

    def psampler(prng, pstate, batch_size, u=None, use_graph=None):
        # jax.pmap(sampler, static_broadcasted_argnums=2)(prng, pstate, batch_size, u) (:236): one engine per local device
        return devices.pmap_sampler(sampler, prng, pstate, batch_size, u, sde, tuple(data_shape), 2, use_graph=use_graph)

    return psampler if is_p else sampler


def get_order0_sampler(sde, model, data_shape, nfe, inverse_scaler, is_em=False, denoising=False, is_p=False):
    num_step = nfe - 1 if denoising else nfe
    rev_ts = get_rev_ts(sde, 2, num_step)  # ts_order is hard-coded to 2 here (cld_jax/sampling.py:162)
    # is_em: Euler step of the probability-flow ODE instead of the exponential integrator (:171-174)
    mean_matrix, eps_matrix = sde.prepare_naive_coef(rev_ts) if is_em else sde.prepare_order0_coef(rev_ts)
    coef = np.stack([mean_matrix, eps_matrix], axis=1)  # n_slots = 2: no epsilon history
    return _make_sampler(sde, model, data_shape, nfe, inverse_scaler, rev_ts, coef, denoising, is_p, "order0")


def _impl_deis_sampler(sde, model, data_shape, nfe, inverse_scaler, deis_order, rev_ts, denoising=False, is_p=False):
    deis_coef = sde.get_deis_coef(deis_order, rev_ts)  # (num_step, deis_order + 3, 2, 2)
    return _make_sampler(sde, model, data_shape, nfe, inverse_scaler, rev_ts, deis_coef, denoising, is_p, "deis")


def get_deis_sampler(sde, model, data_shape, nfe, inverse_scaler, deis_order, ts_order=2, denoising=False, is_p=False):
    rev_ts = get_rev_ts(sde, ts_order, nfe - 1 if denoising else nfe)
    return _impl_deis_sampler(sde, model, data_shape, nfe, inverse_scaler, deis_order, rev_ts, denoising, is_p)


def get_hyd_deis_sampler(sde, model, data_shape, nfe, inverse_scaler, deis_order, noise_nfe_ratio=0.3, img_t_ratio=0.3,
                         ts_order=2.0, denoising=False, is_p=False):
    num_step = nfe - 1 if denoising else nfe
    mid_t = sde.T * img_t_ratio
    noise_nfe = int(num_step * noise_nfe_ratio)
    img_nfe = num_step - noise_nfe
    from .sde_lib import linspace
    ft = np.dtype(getattr(sde, '_ft', np.float64)).type
    noise_ts = linspace(sde.T, mid_t, noise_nfe, endpoint=False, ft=ft)
    # reference: get_rev_ts(sde, ts_order, img_nfe) restarts at sde.T (cld_jax/sampling.py:264) -- kept as is
    img_ts = get_rev_ts(sde, ts_order, img_nfe)
    rev_ts = np.concatenate([noise_ts, img_ts])
    assert rev_ts.shape[0] == num_step + 1
    return _impl_deis_sampler(sde, model, data_shape, nfe, inverse_scaler, deis_order, rev_ts, denoising, is_p)


def _impl_Ldeis_sampler(sde, model, data_shape, nfe, inverse_scaler, deis_order, rev_ts, denoising=False, is_p=False):
    """cld_jax/sampling.py:497-531: eps is mapped by epsR2epsL before every AB step; `sde` is an LSDE."""
    if denoising:
        # the reference builds get_denoising_step(sde=LSDE, ...) here, which reads sde.s_F: the LSDE class has none
        raise AttributeError("'LSDE' object has no attribute 's_F'")
    num_step = nfe
    deis_coef = sde.get_deis_coef(deis_order, rev_ts)
    eps_lin = np.stack([sde.epsR2epsL_matrix(t) for t in rev_ts[:num_step]])
    return _make_sampler(sde, model, data_shape, nfe, inverse_scaler, rev_ts, deis_coef, denoising, is_p, "ldeis", eps_lin=eps_lin,
                         dn_sde=sde.sde)


def get_L_deis_sampler(sde, model, data_shape, nfe, inverse_scaler, deis_order, ts_order=2, denoising=False, is_p=False):
    from .sde_lib import LSDE
    rev_ts = get_rev_ts(sde, ts_order, nfe - 1 if denoising else nfe)
    new_sde = LSDE(sde)
    return _impl_Ldeis_sampler(new_sde, model, data_shape, nfe, inverse_scaler, deis_order, rev_ts, denoising, is_p)


def get_mldeis_sampler(sde, model, data_shape, nfe, inverse_scaler, deis_order, ts_order=2, denoising=False, is_p=False):
    """cld_jax/sampling.py:322-378: the loop runs on y = psi1(t)^-1 x.  As in the reference the denoising step is applied to
    the y-frame state with the x-frame formula, and the final change of frame uses psi1(sampling_eps / 2)."""
    from .sde_lib import MLCLD
    num_step = nfe - 1 if denoising else nfe
    rev_ts = get_rev_ts(sde, ts_order, num_step)
    mlsde = MLCLD(sde)
    deis_coef = mlsde.get_deis_coef(deis_order, rev_ts)
    pre = np.stack([mlsde.y2x_matrix(t) for t in rev_ts[:num_step]])
    return _make_sampler(mlsde, model, data_shape, nfe, inverse_scaler, rev_ts, deis_coef, denoising, is_p, "mldeis", pre=pre,
                         init=mlsde.x2y_matrix(sde.T), post=mlsde.y2x_matrix(sde.sampling_eps / 2), dn_sde=sde)


def _em_coef(sde, rev_ts, lambda_coef):
    """Euler-Maruyama step of cld_jax/sampling.py:625-655 in the [Psi_i, C_i0] form of the fused loop:
    u' = u + (F u - (1+lambda)/2 G G^T score) dt + lambda G xi sqrt|dt|, score = -R^-T eps."""
    n = len(rev_ts) - 1
    coef = np.empty((n, 2, 2, 2))
    noise = np.empty((n, 2, 2))
    for i in range(n):
        t, dt = rev_ts[i], rev_ts[i + 1] - rev_ts[i]
        G = sde.s_G(t)
        coef[i, 0] = np.eye(2) + sde.s_F(t) * dt
        coef[i, 1] = (1.0 + lambda_coef) / 2.0 * (G @ G.T) @ sde.s_invR(t).T * dt
        noise[i] = lambda_coef * G * np.sqrt(abs(dt))
    return coef, noise


def get_em_sampler(sde, model, data_shape, nfe, inverse_scaler, lambda_coef=0, ts_order=2, denoising=False, is_p=False):
    rev_ts = get_rev_ts(sde, ts_order, nfe - 1 if denoising else nfe)
    coef, noise = _em_coef(sde, rev_ts, lambda_coef)
    return _make_sampler(sde, model, data_shape, nfe, inverse_scaler, rev_ts, coef, denoising, is_p, "em",
                         noise=(None, None, noise) if lambda_coef != 0 else None)


def svd_noise_factor(cov):
    """The factor jax.random.multivariate_normal(..., method='svd') multiplies its normal draws with: U sqrt(s) of cov = U s V^T.
    Used for sdeis, whose reference "covariance" is not symmetric (see sde_lib.LambdaSDE): only this rule reproduces its law."""
    u_, s_, _ = np.linalg.svd(cov)
    return u_ * np.sqrt(s_)[None, :]


def _impl_sdeis_update_fn(sde, model, data_shape, nfe, inverse_scaler, deis_order, rev_ts, deis_coef, denoising=False, is_p=False):
    """cld_jax/sampling.py:380-416: AB mean from deis_coef[i][:-1], plus noise of covariance deis_coef[i][-1]."""
    mean_coef = deis_coef[:, :-1]
    noise = np.stack([svd_noise_factor(c) for c in deis_coef[:, -1]])
    return _make_sampler(sde, model, data_shape, nfe, inverse_scaler, rev_ts, mean_coef, denoising, is_p, "sdeis",
                         noise=(None, None, noise), dn_sde=sde.sde)


def _impl_sdeis_sampler(sde, model, data_shape, nfe, inverse_scaler, deis_order, rev_ts, denoising=False, is_p=False):
    deis_coef = np.array(sde.get_deis_coef(deis_order, rev_ts))
    deis_coef[-1, -1] = 0.0  # "avoid numerical error, if we have accurate integral, it should be zero" (:420-421)
    return _impl_sdeis_update_fn(sde, model, data_shape, nfe, inverse_scaler, deis_order, rev_ts, deis_coef, denoising, is_p)


def get_sdeis_sampler(sde, model, data_shape, nfe, inverse_scaler, deis_order, lambda_coef=0, use_order0=True, ts_order=2,
                      denoising=False, is_p=False):
    from .sde_lib import LambdaSDE
    rev_ts = get_rev_ts(sde, ts_order, nfe - 1 if denoising else nfe)
    new_sde = LambdaSDE(sde, lambda_coef, use_order0)
    return _impl_sdeis_sampler(new_sde, model, data_shape, nfe, inverse_scaler, deis_order, rev_ts, denoising, is_p)


def noise_factor(cov):
    """A square root S of a 2x2 covariance (S S^T = cov): Cholesky, falling back to the symmetric root when round-off makes
    the matrix indefinite.  The reference samples multivariate_normal(..., method='svd'); any root gives the same law."""
    try:
        return np.linalg.cholesky(cov)
    except np.linalg.LinAlgError:
        w, v = np.linalg.eigh((cov + cov.T) / 2)
        return (v * np.sqrt(np.clip(w, 0.0, None))) @ v.T


def sscs_ou(sde, s_t, s_t_next):
    """Mean matrix and covariance of the analytic Ornstein-Uhlenbeck half step (cld_jax/sampling.py:542-561)."""
    # (the closed forms below cancel: formed in float64 whatever precision the SDE's tables are in)
    s_t, s_t_next = float(s_t), float(s_t_next)
    beta_int = -1 * (sde.beta_0 * ((1 - s_t_next) - (1 - s_t)) + 0.5 * sde.beta_1 * ((1 - s_t_next) ** 2 - (1 - s_t) ** 2))
    G_ = sde.Gamma
    coeff = np.exp(-2. * beta_int / G_)
    mean_matrix = np.array([[1 + 2 * beta_int / G_, -4 * beta_int / G_ / G_], [beta_int, 1 - 2 * beta_int / G_]]) * coeff
    coeff = np.exp(-4 * beta_int / G_)
    cov_xx = np.exp(4 * beta_int / G_) - 1 - 4 * beta_int / G_ - 8 * beta_int ** 2 / G_ / G_
    cov_xv = -4 * beta_int ** 2 / G_
    cov_vv = (G_ / 2) ** 2 * (np.exp(4 * beta_int / G_) - 1) + beta_int * G_ - 2 * beta_int ** 2
    return mean_matrix, np.array([[cov_xx, cov_xv], [cov_xv, cov_vv]]) * coeff


def get_sscs_sampler(sde, model, data_shape, nfe, inverse_scaler, ts_order=2, denoising=False, is_p=False):
    """Symmetric splitting CLD sampler (cld_jax/sampling.py:563-621): OU half step, score step on v, OU half step.  In the
    fused loop: u <- M1 u + S1 xi before the network, then u <- M2 (A u + B eps) + S2 xi' with A, B the score step."""
    num_step = nfe - 1 if denoising else nfe
    rev_ts = get_rev_ts(sde, ts_order, num_step)
    ts = 1 - rev_ts
    coef = np.empty((num_step, 2, 2, 2))
    pre_lin, pre_noise, post_noise = (np.empty((num_step, 2, 2)) for _ in range(3))
    for i in range(num_step):
        cur_t, next_t = ts[i], ts[i + 1]
        mid = (cur_t + next_t) / 2.0
        M1, cov1 = sscs_ou(sde, cur_t, mid)
        M2, cov2 = sscs_ou(sde, mid, next_t)
        k = 2 * float(sde.beta(cur_t)) * sde.Gamma * (next_t - cur_t)
        A = np.eye(2) + np.array([[0.0, 0.0], [0.0, k * sde.m_inv]])
        RinvT = sde.s_invR(sde.T - cur_t).T  # score = -R^-T eps at time T - s_t; only its v component enters
        B = k * np.array([[0.0, 0.0], [-RinvT[1, 0], -RinvT[1, 1]]])
        coef[i, 0], coef[i, 1] = M2 @ A, M2 @ B
        pre_lin[i], pre_noise[i], post_noise[i] = M1, noise_factor(cov1), noise_factor(cov2)
    return _make_sampler(sde, model, data_shape, nfe, inverse_scaler, rev_ts, coef, denoising, is_p, "sscs",
                         noise=(pre_lin, pre_noise, post_noise))


def get_ode_sampler(sde, model, data_shape, inverse_scaler, denoising=False, rtol=1e-5, atol=1e-5, method='RK45', is_p=False):
    """Adaptive probability-flow ODE solve (cld_jax/sampling.py:432-495): scipy's solve_ivp on the host drives the drift
    F u - 1/2 G G score evaluated on the GPU.  As in the reference, the pmapped variant (is_p=True, what get_sampling_fn
    builds) divides batch_size by the device count and skips the denoising step."""
    from scipy import integrate
    step_fn = get_denoising_step(sde, model, sde.sampling_eps, sde.sampling_eps) if denoising else (lambda _, u: u)

    def _solve(state, u, slot=0):
        import torch
        score_fn = mutils.get_score_fn(sde, model, state.params_ema, state.model_state, train=False, continuous=True, slot=slot)
        d_shape = tuple(u.shape)
        ones = torch.ones(d_shape[0], device=u.device, dtype=torch.float32)

        def ode_func(t, x):
            xt = torch.as_tensor(np.asarray(x, dtype=np.float32).reshape(d_shape), device=u.device)
            score = score_fn(xt, ones * float(t))
            cur_F, cur_G = sde.s_F(t), sde.s_G(t)
            grad = utils.sbmm(cur_F, xt) - 0.5 * utils.sbmm(cur_G @ cur_G, score)
            return grad.reshape(-1).double().cpu().numpy()

        solution = integrate.solve_ivp(ode_func, (sde.T, sde.sampling_eps), u.reshape(-1).double().cpu().numpy(), rtol=rtol,
                                       atol=atol, method=method)
        out = torch.as_tensor(solution.y[:, -1].reshape(d_shape), device=u.device, dtype=torch.float32)
        return out, solution.nfev

    def _prep(rng, u, shape):
        import torch
        if u is None:
            u = sde.prior_sampling(_seed_of(rng), shape)
        if not hasattr(u, 'is_cuda'):
            u = torch.as_tensor(np.asarray(u), dtype=torch.float32)
        if not u.is_cuda:
            u = u.to(torch.device('cuda', torch.cuda.current_device()))
        return u.to(torch.float32)

    def sampler(rng, state, batch_size, u=None):
        u = _prep(rng, u, (batch_size,) + tuple(data_shape))
        u, nfe = _solve(state, u)
        u = step_fn(state, u)
        x, v = u[..., 0], u[..., 1]
        return inverse_scaler(x), v, nfe

    def psampler(prng, pstate, batch_size, u=None):
        # as in the reference (:480-493) the pmapped variant divides batch_size by the device count once more; the adaptive solve is
        # driven by the host (scipy), so the devices take turns
        devs = devices.local_devices()
        batch_size = batch_size // len(devs)

        def one(rng, state, bs, ud, slot=0):
            out, nfe = _solve(state, ud, slot)
            return inverse_scaler(out[..., 0]), out[..., 1], nfe

        if u is None:
            import torch
            seeds = devices.device_seeds(prng, len(devs))
            u = [_prep(seeds[d], None, (batch_size,) + tuple(data_shape)).to(torch.device('cuda', devs[d])) for d in range(len(devs))]
        return devices.pmap_sampler(one, prng, pstate, batch_size, u, sde, tuple(data_shape) + (2,), 2)

    return psampler if is_p else sampler


def get_sampling_fn(config, sde, model, shape, inverse_scaler):
    """Create a sampling function (cld_jax/sampling.py:41-154)."""
    del shape
    sampler_name = config.sampling.method
    data_shape = utils.get_data_shape(config)
    name = sampler_name.lower()
    if name == 'order0':
        return get_order0_sampler(sde=sde, model=model, data_shape=data_shape, nfe=config.sampling.nfe,
                                  inverse_scaler=inverse_scaler, is_em=config.sampling.is_em,
                                  denoising=config.sampling.noise_removal, is_p=True)
    if name == 'deis':
        return get_deis_sampler(sde=sde, model=model, data_shape=data_shape, nfe=config.sampling.nfe,
                                inverse_scaler=inverse_scaler, deis_order=config.sampling.deis_order,
                                ts_order=config.sampling.ts_order, denoising=config.sampling.noise_removal, is_p=True)
    if name == 'hybdeis':
        return get_hyd_deis_sampler(sde=sde, model=model, data_shape=data_shape, nfe=config.sampling.nfe,
                                    inverse_scaler=inverse_scaler, deis_order=config.sampling.deis_order,
                                    noise_nfe_ratio=config.sampling.noise_nfe_ratio,
                                    img_t_ratio=config.sampling.img_t_ratio, ts_order=config.sampling.ts_order,
                                    denoising=config.sampling.noise_removal, is_p=True)
    if name == 'ldeis':
        return get_L_deis_sampler(sde=sde, model=model, data_shape=data_shape, nfe=config.sampling.nfe,
                                  inverse_scaler=inverse_scaler, deis_order=config.sampling.deis_order,
                                  ts_order=config.sampling.ts_order, denoising=config.sampling.noise_removal, is_p=True)
    if name == 'mldeis':
        return get_mldeis_sampler(sde=sde, model=model, data_shape=data_shape, nfe=config.sampling.nfe,
                                  inverse_scaler=inverse_scaler, deis_order=config.sampling.deis_order,
                                  ts_order=config.sampling.ts_order, denoising=config.sampling.noise_removal, is_p=True)
    if name == 'ode':
        return get_ode_sampler(sde=sde, model=model, data_shape=data_shape, inverse_scaler=inverse_scaler,
                               denoising=config.sampling.noise_removal, atol=config.sampling.atol, rtol=config.sampling.rtol,
                               method=config.sampling.ode_method, is_p=True)
    if name == 'em':
        return get_em_sampler(sde=sde, model=model, data_shape=data_shape, nfe=config.sampling.nfe, inverse_scaler=inverse_scaler,
                              lambda_coef=config.sampling.lambda_coef, ts_order=config.sampling.ts_order,
                              denoising=config.sampling.noise_removal, is_p=True)
    if name == 'sscs':
        return get_sscs_sampler(sde=sde, model=model, data_shape=data_shape, nfe=config.sampling.nfe, inverse_scaler=inverse_scaler,
                                ts_order=config.sampling.ts_order, denoising=config.sampling.noise_removal, is_p=True)
    if name == 'sdeis':
        return get_sdeis_sampler(sde=sde, model=model, data_shape=data_shape, nfe=config.sampling.nfe, inverse_scaler=inverse_scaler,
                                 deis_order=config.sampling.deis_order, lambda_coef=config.sampling.lambda_coef,
                                 use_order0=config.sampling.sdeis_use_order0, ts_order=config.sampling.ts_order,
                                 denoising=config.sampling.noise_removal, is_p=True)
    raise RuntimeError
