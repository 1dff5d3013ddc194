"""ORACLE (test infrastructure only): the CLD sampler variants that share the DEIS loop.

Restates, in numpy float64, deliberately literal (python loops):

  /root/reference/cld_jax/sde_lib.py:121-180   s_f1_psi, s_psi1, s_inv_psi1, s_F1, s_F2
  /root/reference/cld_jax/sde_lib.py:469-520   class LSDE (s_L, epsR2epsL, s_eps_integrand, get_deis_coef)
  /root/reference/cld_jax/sampling.py:272-320  get_ml_psi2_fn, class MLCLD
  /root/reference/cld_jax/sampling.py:322-378  get_mldeis_sampler
  /root/reference/cld_jax/sampling.py:497-540  _impl_Ldeis_sampler, get_L_deis_sampler
  /root/reference/cld_jax/sampling.py:255-269  get_hyd_deis_sampler (the two-phase time grid of `hybdeis`)
  /root/reference/cld_jax/models/utils.py:168-182  get_eps_fn with sde.mixed_score=True
  /root/reference/cld_jax/sampling.py:432-463  get_ode_sampler (the non-pmapped `sampler`)
  /root/reference/cld_jax/sampling.py:625-655  get_em_sampler (noise injected by the caller: jax's threefry stream is not
                                               reproducible, see oracle/__init__.py)
  /root/reference/cld_jax/sampling.py:542-621  get_sscs_ou_fn, get_sscs_score_fn, get_sscs_sampler (noise injected likewise;
                                               multivariate_normal(method="svd") is replaced by a Cholesky root: same law)

PARITY UNPINNED -- see oracle/__init__.py.  What pins these restatements without the reference: psi2(t) against the closed
form psi1(t)^-1 expm(int_0^t F) (the two frames describe the same flow), L L^T = R R^T, and agreement of the three
samplers with each other in the NFE -> infinity limit on an analytic eps (tests/test_oracle_cld_ext.py).
"""
import numpy as np

from . import cld as ocld


def s_f1_psi(sde, s, t):
    # sde_lib.py:121-143 (the un-commented return)
    beta_int = sde._beta_int(t) - sde._beta_int(s)
    sqrt_m = np.sqrt(1.0 / sde.m_inv)
    inv_sqrt_m = np.sqrt(sde.m_inv)
    return np.asarray([[np.cos(beta_int * inv_sqrt_m), inv_sqrt_m * np.sin(beta_int * inv_sqrt_m)],
                       [-sqrt_m * np.sin(beta_int * inv_sqrt_m), np.cos(beta_int * inv_sqrt_m)]], dtype=np.float64)


def s_psi1(sde, t):
    return s_f1_psi(sde, 0.0, t)  # :146-149


def s_inv_psi1(sde, t):
    return s_f1_psi(sde, t, 0.0)  # :152-155


def s_F2(sde, t):
    cur_beta = sde._beta(t)  # :169-179
    return np.asarray([[0.0, 0.0], [0.0, -sde.Gamma * cur_beta * sde.m_inv]], dtype=np.float64)


def runge_kutta(x, t, dt, fn):
    # deis.py:6-17
    grad_1 = fn(x, t)
    x_2 = x + grad_1 * dt / 2
    grad_2 = fn(x_2, t + dt / 2)
    x_3 = x + grad_2 * dt / 2
    grad_3 = fn(x_3, t + dt / 2)
    x_4 = x + grad_3 * dt
    grad_4 = fn(x_4, t + dt)
    return x + dt / 6 * (grad_1 + 2 * grad_2 + 2 * grad_3 + grad_4)


def mixed_eps_fn(sde, eps_fn):
    """get_eps_fn with sde.mixed_score (models/utils.py:174-176): + bmm(v_invR(t), x.at[..., 0].set(0))."""

    def fn(x, t):
        out = eps_fn(x, t)
        x0 = x.copy()
        x0[..., 0] = 0.0
        inv_r = np.stack([ocld.inv_2x2(sde.s_R(np.float64(tb))) for tb in t])
        return out + ocld.bmm(inv_r, x0)

    return fn


class LSDE:
    # sde_lib.py:469-520
    def __init__(self, sde):
        self.sde = sde
        self.dtype = sde.dtype
        self.T, self.sampling_eps = sde.T, sde.sampling_eps
        self.s_G = sde.s_G

    def s_L(self, s_t):
        R = self.sde.s_R(np.float64(s_t))
        return np.linalg.cholesky(R @ R.T)  # :489-492

    def epsR2epsL(self, s_t, eps):
        cur_L = self.s_L(s_t)
        cur_R = self.sde.s_R(np.float64(s_t))
        coef = cur_L.T @ ocld.inv_2x2(cur_R.T)  # :494-500
        return ocld.sbmm(coef, eps)

    def s_eps_integrand(self, s_t):
        cur_G = self.s_G(s_t)
        return 0.5 * cur_G @ cur_G @ ocld.inv_2x2(self.s_L(s_t)).T  # :503-507

    def vs_psi(self, s_vec, t):
        return ocld.vs_psi(self.sde, s_vec, t)

    def v_eps_integrand(self, t_vec):
        # vmap(s_eps_integrand), batched (tests check it against the scalar form above)
        Rs = self.sde.v_R(t_vec)
        Ls = np.linalg.cholesky(Rs @ np.swapaxes(Rs, 1, 2))
        invLT = np.swapaxes(np.linalg.inv(Ls), 1, 2)
        g2 = 2 * self.sde.Gamma * self.sde._beta(np.asarray(t_vec, dtype=np.float64))
        out = np.zeros_like(Rs)
        out[:, 1, :] = 0.5 * g2[:, None] * invLT[:, 1, :]
        return out

    def get_deis_coef(self, order, rev_timesteps):
        rev_timesteps = np.asarray(rev_timesteps, dtype=np.float64)
        x_coef = np.stack([self.sde.s_psi(rev_timesteps[i], rev_timesteps[i + 1]) for i in range(len(rev_timesteps) - 1)])
        eps_coef = ocld.get_ab_eps_coef(self, order + 1, rev_timesteps, order)
        return np.concatenate([x_coef[:, None], eps_coef], axis=1)


_PSI2 = {}


def get_ml_psi2_fn(sde, N=100000):
    # sampling.py:272-285: scan of runge_kutta over N + 1 steps of size 1/N from (eye(2), 0.0), recording the carry
    key = (float(sde.m_inv), float(sde.beta_0), float(sde.beta_1), N)
    if key not in _PSI2:
        def fn(psi2, cur_t):
            return s_inv_psi1(sde, cur_t) @ s_F2(sde, cur_t) @ s_psi1(sde, cur_t) @ psi2
        psi2, cur_t = np.eye(2), 0.0
        psi2s = np.empty((N + 1, 2, 2))
        ts = np.empty(N + 1)
        dt = 1 / N
        for k in range(N + 1):
            psi2s[k], ts[k] = psi2, cur_t
            psi2 = runge_kutta(psi2, cur_t, dt, fn)
            cur_t = cur_t + dt
        _PSI2[key] = ocld.get_interp_fn(ts, psi2s)
    return _PSI2[key]


class MLCLD:
    # sampling.py:287-320
    def __init__(self, sde, N=100000):
        assert sde.beta_1 == 0
        self.sde = sde
        self.dtype = sde.dtype
        self.s_psi2_fn = get_ml_psi2_fn(sde, N)
        self.sampling_eps, self.T = sde.sampling_eps, sde.T
        self.s_G = sde.s_G

    def y2x(self, y, t):
        return ocld.sbmm(s_psi1(self.sde, t), y)

    def x2y(self, x, t):
        return ocld.sbmm(s_inv_psi1(self.sde, t), x)

    def s_psi(self, s, t):
        return self.s_psi2_fn(np.float64(t)) @ np.linalg.inv(self.s_psi2_fn(np.float64(s)))

    def s_eps_integrand(self, s_t):
        cur_G = self.s_G(s_t)
        cur_inv_psi1 = s_inv_psi1(self.sde, s_t)
        return 0.5 * cur_inv_psi1 @ cur_G @ cur_G.T @ ocld.inv_2x2(self.sde.s_R(np.float64(s_t))).T

    def vs_psi(self, s_vec, t):
        # vmap(s_psi, (0, None)), batched (tests check it against the scalar form above)
        return self.s_psi2_fn(np.float64(t)) @ np.linalg.inv(self.s_psi2_fn(np.asarray(s_vec, dtype=np.float64)))

    def v_eps_integrand(self, t_vec):
        t_vec = np.asarray(t_vec, dtype=np.float64)
        b = self.sde._beta_int(t_vec)
        sqrt_m, inv_sqrt_m = np.sqrt(1.0 / self.sde.m_inv), np.sqrt(self.sde.m_inv)
        ip1 = np.empty((len(t_vec), 2, 2))  # s_f1_psi(t, 0): beta_int changes sign
        ip1[:, 0, 0], ip1[:, 0, 1] = np.cos(-b * inv_sqrt_m), inv_sqrt_m * np.sin(-b * inv_sqrt_m)
        ip1[:, 1, 0], ip1[:, 1, 1] = -sqrt_m * np.sin(-b * inv_sqrt_m), np.cos(-b * inv_sqrt_m)
        return ip1 @ ocld.v_eps_integrand(self.sde, t_vec)  # G G^T = G G for the diagonal G

    def get_deis_coef(self, order, rev_timesteps):
        rev_timesteps = np.asarray(rev_timesteps, dtype=np.float64)
        x_coef = np.stack([self.s_psi(rev_timesteps[i], rev_timesteps[i + 1]) for i in range(len(rev_timesteps) - 1)])
        eps_coef = ocld.get_ab_eps_coef(self, order + 1, rev_timesteps, order)
        return np.concatenate([x_coef[:, None], eps_coef], axis=1)


def hyd_rev_ts(sde, nfe, noise_nfe_ratio=0.3, img_t_ratio=0.3, ts_order=2.0, denoising=False):
    """The time grid of get_hyd_deis_sampler (sampling.py:259-268): int(num_step * noise_nfe_ratio) uniform knots on
    [T, T * img_t_ratio) followed by get_rev_ts(sde, ts_order, img_nfe) -- which starts again at T (:264), kept literally."""
    num_step = nfe - 1 if denoising else nfe
    mid_t = sde.T * img_t_ratio
    noise_nfe = int(num_step * noise_nfe_ratio)
    img_nfe = num_step - noise_nfe
    noise_ts = ocld.jnp_linspace(sde.T, mid_t, noise_nfe, endpoint=False, dtype=sde.dtype)
    img_ts = ocld.get_rev_ts(sde, ts_order, img_nfe)
    rev_ts = np.concatenate([noise_ts, img_ts])
    assert rev_ts.shape[0] == num_step + 1  # :268
    return rev_ts


def hybdeis_sampler(sde, eps_fn, u, nfe, deis_order, noise_nfe_ratio=0.3, img_t_ratio=0.3, ts_order=2.0, denoising=False,
                    inverse_scaler=None, deis_coef=None):
    """get_hyd_deis_sampler (sampling.py:255-269): _impl_deis_sampler on the two-phase grid."""
    rev_ts = hyd_rev_ts(sde, nfe, noise_nfe_ratio, img_t_ratio, ts_order, denoising)
    return ocld.deis_sampler(sde, eps_fn, u, nfe, deis_order, denoising=denoising, inverse_scaler=inverse_scaler,
                             deis_coef=deis_coef, rev_ts=rev_ts)


def ldeis_sampler(sde, eps_fn, u, nfe, deis_order, ts_order=2, inverse_scaler=None, deis_coef=None):
    """get_L_deis_sampler with denoising=False (with denoising the reference raises: LSDE has no s_F)."""
    rev_ts = ocld.get_rev_ts(sde, ts_order, nfe)
    new_sde = LSDE(sde)
    if deis_coef is None:
        deis_coef = new_sde.get_deis_coef(deis_order, rev_ts)
    ones = np.ones(u.shape[0], dtype=u.dtype)
    eps_pred = np.stack([u] * (deis_order + 1))
    for i in range(nfe):
        cur_eps_pred = eps_fn(u, ones * u.dtype.type(rev_ts[i]))
        cur_eps_pred = new_sde.epsR2epsL(rev_ts[i], cur_eps_pred).astype(u.dtype)
        u, eps_pred = ocld.multistep_ab_step(u, deis_coef[i].astype(u.dtype), cur_eps_pred, eps_pred)
    x, v = u[..., 0], u[..., 1]
    if inverse_scaler is not None:
        x = inverse_scaler(x)
    return x, v, nfe


def mldeis_sampler(sde, eps_fn, u, nfe, deis_order, ts_order=2, denoising=False, inverse_scaler=None, deis_coef=None,
                   mlsde=None):
    """get_mldeis_sampler (sampling.py:322-378), non-pmapped."""
    num_step = nfe - 1 if denoising else nfe
    rev_ts = ocld.get_rev_ts(sde, ts_order, num_step)  # sampling.py:330-337 repeats get_rev_ts inline
    mlsde = MLCLD(sde) if mlsde is None else mlsde
    if deis_coef is None:
        deis_coef = mlsde.get_deis_coef(deis_order, rev_ts)
    u = mlsde.x2y(u, sde.T).astype(u.dtype)
    ones = np.ones(u.shape[0], dtype=u.dtype)

    def y_eps_fn(y_u, s_t):
        x_u = mlsde.y2x(y_u, s_t).astype(y_u.dtype)
        return eps_fn(x_u, s_t * ones)

    eps_pred = np.stack([u] * (deis_order + 1))
    for i in range(num_step):
        cur_eps_pred = y_eps_fn(u, u.dtype.type(rev_ts[i]))
        u, eps_pred = ocld.multistep_ab_step(u, deis_coef[i].astype(u.dtype), cur_eps_pred, eps_pred)
    if denoising:
        u = ocld.get_denoising_step(sde, eps_fn, sde.sampling_eps, sde.sampling_eps)(u)
    u = mlsde.y2x(u, sde.sampling_eps / 2).astype(u.dtype)
    x, v = u[..., 0], u[..., 1]
    if inverse_scaler is not None:
        x = inverse_scaler(x)
    return x, v, nfe


def ode_sampler(sde, eps_fn, u, denoising=False, rtol=1e-5, atol=1e-5, method="RK45", inverse_scaler=None):
    """get_ode_sampler.sampler (sampling.py:437-463)."""
    from scipy import integrate
    d_shape = u.shape
    ones = np.ones(u.shape[0], dtype=u.dtype)

    def drift_fn(x, s_t):
        vec_t = ones * s_t
        score = sde.eps2score(eps_fn(x, vec_t), vec_t)
        cur_F, cur_G = sde.s_F(s_t), sde.s_G(s_t)
        return ocld.sbmm(cur_F, x) - 0.5 * ocld.sbmm(cur_G @ cur_G, score)

    def ode_func(t, x):
        return drift_fn(x.reshape(d_shape), t).reshape(-1)

    solution = integrate.solve_ivp(ode_func, (sde.T, sde.sampling_eps), u.reshape(-1), rtol=rtol, atol=atol, method=method)
    nfe = solution.nfev
    u = solution.y[:, -1].reshape(d_shape)
    if denoising:
        u = ocld.get_denoising_step(sde, eps_fn, sde.sampling_eps, sde.sampling_eps)(u)
    x, v = u[..., 0], u[..., 1]
    if inverse_scaler is not None:
        x = inverse_scaler(x)
    return x, v, nfe


def em_sampler(sde, eps_fn, u, nfe, lambda_coef=0, ts_order=2, denoising=False, inverse_scaler=None, noise_fn=None):
    """get_em_sampler.sampler (sampling.py:633-655); noise_fn(draw, shape) stands for random.normal(rng_i, shape), draw = 2 i + 1
    for step i (the engine's numbering: even draws belong to pre-network updates, which this sampler has none of)."""
    num_step = nfe - 1 if denoising else nfe
    rev_ts = ocld.get_rev_ts(sde, ts_order, num_step)
    ones = np.ones(u.shape[0], dtype=u.dtype)
    for i in range(num_step):
        cur_t, next_t = rev_ts[i], rev_ts[i + 1]
        delta_t = next_t - cur_t
        cur_G = sde.s_G(cur_t)
        cur_score = sde.eps2score(eps_fn(u, cur_t * ones), cur_t * ones)
        grad = ocld.sbmm(sde.s_F(cur_t), u) - (1.0 + lambda_coef) / 2.0 * ocld.sbmm(cur_G @ cur_G.T, cur_score)
        noise = (noise_fn(2 * i + 1, u.shape) if lambda_coef != 0 else np.zeros_like(u)) * np.sqrt(np.abs(delta_t))
        u = u + grad * delta_t + ocld.sbmm(cur_G, noise) * lambda_coef
    if denoising:
        u = ocld.get_denoising_step(sde, eps_fn, sde.sampling_eps, sde.sampling_eps)(u)
    x, v = u[..., 0], u[..., 1]
    if inverse_scaler is not None:
        x = inverse_scaler(x)
    return x, v, nfe


def sscs_ou(sde, u, s_t, s_t_next, xi):
    # get_sscs_ou_fn._fn (sampling.py:543-560); xi: standard normal pairs standing for the multivariate_normal draw
    beta_int = sde._beta_int(1 - s_t_next) - sde._beta_int(1 - s_t)
    beta_int = -1 * beta_int
    coeff = np.exp(-2. * beta_int / sde.Gamma)
    mean_matrix = np.asarray([[1 + 2 * beta_int / sde.Gamma, -4 * beta_int / sde.Gamma / sde.Gamma],
                              [beta_int, 1 - 2 * beta_int / sde.Gamma]]) * coeff
    mean = ocld.sbmm(mean_matrix, u)
    coeff = np.exp(-4 * beta_int / sde.Gamma)
    cov_xx = np.exp(4 * beta_int / sde.Gamma) - 1 - 4 * beta_int / sde.Gamma - 8 * beta_int ** 2 / sde.Gamma / sde.Gamma
    cov_xv = -4 * beta_int ** 2 / sde.Gamma
    cov_vv = (sde.Gamma / 2) ** 2 * (np.exp(4 * beta_int / sde.Gamma) - 1) + beta_int * sde.Gamma - 2 * beta_int ** 2
    cov = np.asarray([[cov_xx, cov_xv], [cov_xv, cov_vv]]) * coeff
    return mean + ocld.sbmm(np.linalg.cholesky(cov), xi), cov


def sscs_sampler(sde, eps_fn, u, nfe, ts_order=2, denoising=False, inverse_scaler=None, noise_fn=None):
    """get_sscs_sampler.sampler (sampling.py:573-600); noise_fn(draw, shape): standard normals, draws 2 i (first OU half step)
    and 2 i + 1 (second) of step i."""
    num_step = nfe - 1 if denoising else nfe
    rev_ts = ocld.get_rev_ts(sde, ts_order, num_step)
    ts = 1 - rev_ts
    ones = np.ones(u.shape[0], dtype=u.dtype)
    for i in range(num_step):
        cur_t, next_t = ts[i], ts[i + 1]
        u, _ = sscs_ou(sde, u, cur_t, (cur_t + next_t) / 2.0, noise_fn(2 * i, u.shape))
        # get_sscs_score_fn._fn (:564-572)
        x, v = u[..., 0], u[..., 1]
        t_net = (sde.T - cur_t) * ones
        v_score = sde.eps2score(eps_fn(u, t_net), t_net)[..., 1]
        v_dot = 2 * sde._beta(cur_t) * sde.Gamma * (v_score + sde.m_inv * v)
        v = v + v_dot * (next_t - cur_t)
        u = np.stack([x, v], axis=-1)
        u, _ = sscs_ou(sde, u, (cur_t + next_t) / 2.0, next_t, noise_fn(2 * i + 1, u.shape))
    if denoising:
        u = ocld.get_denoising_step(sde, eps_fn, sde.sampling_eps, sde.sampling_eps)(u)
    x, v = u[..., 0], u[..., 1]
    if inverse_scaler is not None:
        x = inverse_scaler(x)
    return x, v, nfe


class LambdaSDE:
    """sde_lib.py:334-466: the lambda-family of reverse SDEs (lambda = 0: probability flow; 1: the reverse diffusion)."""

    def __init__(self, sde, lambda_coef=0.1, use_order0=True, dt=1e-5):
        self.sde = sde
        self.dtype = sde.dtype
        self.use_order0 = use_order0
        self.lambda_coef = lambda_coef
        self.T, self.sampling_eps = sde.T, sde.sampling_eps
        self.s_hat_psi_02t = self._get_hat_psi_02t(dt)

    def s_cov(self, t):
        R = self.sde.s_R(np.float64(t))
        return R @ R.T  # sde_lib.py:242-244

    def s_hat_F(self, t):
        cur_G = self.sde.s_G(t)
        inv_cov = ocld.inv_2x2(self.s_cov(t))
        return self.sde.s_F(t) + 0.5 * (1 + self.lambda_coef ** 2) * cur_G @ cur_G.T @ inv_cov  # :351-355

    def _get_hat_psi_02t(self, dt):
        # :357-376: scan of runge_kutta(step dt) over linspace(0, 1 + dt, int(1/dt) + 1, endpoint=False), emitting the carry
        n = int(1.0 / dt) + 1
        ts = np.linspace(0, 1.0 + dt, n, endpoint=False)
        Rs = np.empty((n, 2, 2))
        x = np.eye(2)
        fn = lambda _x, _t: self.s_hat_F(_t) @ _x
        for k in range(n):
            Rs[k] = x
            x = runge_kutta(x, ts[k], dt, fn)
        return ocld.get_interp_fn(ts, Rs)

    def s_hat_psi(self, s, t):
        return self.s_hat_psi_02t(np.float64(t)) @ ocld.inv_2x2(self.s_hat_psi_02t(np.float64(s)))  # :378-380

    def cond_rev_cov(self, s, t, n_step=10_000):
        # :382-399 (as coded: `_x @ cur_hat_F` without the transpose of the comment above it, and a time grid of n_step + 1
        # points for n_step steps of size (t - s) / n_step)
        dir_sign = 1 if t > s else -1
        dt = (t - s) * 1.0 / n_step
        ts = np.linspace(s, t, n_step + 1, endpoint=False)

        def _ode_fn(_x, _t):
            cur_hat_F = self.s_hat_F(_t)
            cur_G = self.sde.s_G(_t)
            return cur_hat_F @ _x + _x @ cur_hat_F + dir_sign * self.lambda_coef ** 2 * cur_G @ cur_G.T

        val = np.zeros((2, 2))
        for i in range(n_step):
            val = runge_kutta(val, ts[i], dt, _ode_fn)
        return val

    def update_coef(self, s, t):
        x_coef = self.sde.s_psi(s, t)
        eps_coef = (self.s_hat_psi(s, t) - x_coef) @ self.sde.s_R(np.float64(s))
        return np.stack([x_coef, eps_coef, self.cond_rev_cov(s, t)])  # :401-407

    def get_poly_eps_coef(self, order, rev_timesteps):
        outer = self

        class _sde:  # :409-424
            dtype = self.dtype

            @staticmethod
            def vs_psi(ss, t):
                return np.stack([outer.s_hat_psi(s, t) for s in ss])

            @staticmethod
            def v_eps_integrand(ts):
                out = []
                for _t in ts:
                    cur_G = outer.sde.s_G(_t)
                    inv_cov = ocld.inv_2x2(outer.s_cov(_t))
                    out.append(0.5 * (1 + outer.lambda_coef ** 2) * cur_G @ cur_G.T @ inv_cov @ outer.sde.s_psi(0, _t))
                return np.stack(out)

        ab_eps_coef = ocld.get_ab_eps_coef(_sde, order + 1, rev_timesteps, order)
        last_term = np.stack([self.sde.s_psi(s, 0) @ self.sde.s_R(np.float64(s)) for s in rev_timesteps[:-1]])  # :426-432
        return np.einsum("b...ij,bjk->b...ik", ab_eps_coef, last_term)

    def get_order0_coef(self, rev_timesteps):
        return np.stack([self.update_coef(rev_timesteps[i], rev_timesteps[i + 1]) for i in range(len(rev_timesteps) - 1)])

    def get_deis_coef(self, order, rev_timesteps):
        rev_timesteps = np.asarray(rev_timesteps, dtype=np.float64)
        if self.use_order0 and order == 0:
            c = self.get_order0_coef(rev_timesteps)
            return np.stack([c[:, 0], c[:, 1], np.zeros(c[:, 0].shape), c[:, 2]], axis=1)  # :436-445
        x_coef = np.stack([self.sde.s_psi(rev_timesteps[i], rev_timesteps[i + 1]) for i in range(len(rev_timesteps) - 1)])
        eps_coef = self.get_poly_eps_coef(order, rev_timesteps)
        covs = np.stack([self.cond_rev_cov(rev_timesteps[i], rev_timesteps[i + 1]) for i in range(len(rev_timesteps) - 1)])
        return np.concatenate([x_coef[:, None], eps_coef, covs[:, None]], axis=1)  # :451-454


def sdeis_sampler(sde, eps_fn, u, nfe, deis_order, lambda_coef=0, use_order0=True, ts_order=2, denoising=False, inverse_scaler=None,
                  noise_fn=None, new_sde=None, deis_coef=None):
    """get_sdeis_sampler -> _impl_sdeis_sampler -> _impl_sdeis_update_fn.sampler (sampling.py:380-427).  noise_fn(draw, shape):
    standard normals for step i at draw 2 i + 1.  The reference samples random.multivariate_normal(rng, 0, cov, method="svd"), whose
    published definition is factor = U sqrt(s) for cov = U s V^T; cond_rev_cov as coded is not symmetric, so only that rule (not a
    Cholesky root) reproduces the law of the reference's draws."""
    num_step = nfe - 1 if denoising else nfe
    rev_ts = ocld.get_rev_ts(sde, ts_order, num_step)
    if deis_coef is None:
        new_sde = LambdaSDE(sde, lambda_coef, use_order0) if new_sde is None else new_sde
        deis_coef = new_sde.get_deis_coef(deis_order, rev_ts)
    deis_coef = deis_coef.copy()
    deis_coef[-1, -1] = 0.0  # :420-421
    ones = np.ones(u.shape[0], dtype=u.dtype)
    eps_pred = np.stack([u] * (deis_order + 1))
    for i in range(num_step):
        cur_eps_pred = eps_fn(u, ones * u.dtype.type(rev_ts[i]))
        mean, eps_pred = ocld.multistep_ab_step(u, deis_coef[i][:-1].astype(u.dtype), cur_eps_pred, eps_pred)
        u = mean + ocld.sbmm(svd_factor(deis_coef[i][-1]), noise_fn(2 * i + 1, mean.shape))
    if denoising:
        u = ocld.get_denoising_step(sde, eps_fn, sde.sampling_eps, sde.sampling_eps)(u)
    x, v = u[..., 0], u[..., 1]
    if inverse_scaler is not None:
        x = inverse_scaler(x)
    return x, v, nfe


def svd_factor(cov):
    u_, s_, _ = np.linalg.svd(cov)
    return u_ * np.sqrt(s_)[None, :]


def noise_root(cov):
    """S with S S^T = cov: Cholesky, or the symmetric root when round-off leaves cov indefinite (e.g. the zeroed last step)."""
    try:
        return np.linalg.cholesky(cov)
    except np.linalg.LinAlgError:
        w, v = np.linalg.eigh((cov + cov.T) / 2)
        return (v * np.sqrt(np.clip(w, 0.0, None))) @ v.T
