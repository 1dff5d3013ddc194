"""Critically-damped Langevin diffusion: SDE algebra for the sampler (host side, numpy).

Replaces the sampling members of cld_jax/sde_lib.py:45-331 (class CLD, from_config).

Arithmetic.  The reference computes every table in the dtype jax is configured with and asserts that it matches
`config.model.x64` (sde_lib.py:112,314): float32 unless x64 is set -- and x64=False is its default
(configs/default_cifar10_config.py).  `CLD(x64=...)` is honoured the same way: with x64=False the R(t) scan (native
`gd_cld_r_table_f32`), its interpolation, Psi, the epsilon integrand, the time grids and the 10 000-term Riemann sums of the
DEIS table are evaluated in float32, operation by operation in the order the reference's expressions are written; with
x64=True the same code runs in float64.  What cannot be reproduced is XLA's summation order inside `jnp.sum` (here: numpy's
pairwise sum) and its libm (exp, sqrt, pow): DESIGN.md section 2 prices both.  The sampler variants further down (LSDE,
MLCLD, LambdaSDE) read R(t) from this class in the configured precision and build their own tables in float64.
"""
import ctypes

import numpy as np

from .. import _abi
from . import deis


def inv_2x2(m):
    a, b, c, d = m[..., 0, 0], m[..., 0, 1], m[..., 1, 0], m[..., 1, 1]
    coef = 1.0 / (a * d - b * c)
    out = np.empty_like(m)
    out[..., 0, 0], out[..., 0, 1], out[..., 1, 0], out[..., 1, 1] = d * coef, -b * coef, -c * coef, a * coef
    return out


def linspace(start, stop, num, endpoint=True, ft=np.float64):
    """jnp.linspace for a floating dtype as jax 0.2.8 evaluates it (jax/_src/numpy/lax_numpy.py, from memory: the form that
    "recovers the endpoints"): start * (1 - step) + stop * step with step = iota / div, all in the computation dtype."""
    ft = np.dtype(ft).type
    div = (num - 1) if endpoint else num
    if num == 1:
        return np.asarray([start], dtype=ft)
    step = np.arange(num, dtype=ft) / ft(div)
    return ft(start) * (ft(1) - step) + ft(stop) * step


def linspace_int(start, stop, num, ft=np.float64):
    """jnp.linspace(..., dtype=int): floor(start + iota * delta) with delta = (stop - start) / (num - 1) in the floating
    computation dtype (float32 unless x64) -- in float32 some products round up across an integer (i = 9 999 of the
    R-table knots: 99 991 instead of 99 990), so the knot rows are the reference's only if the same arithmetic picks them."""
    ft = np.dtype(ft).type
    delta = (ft(stop) - ft(start)) / ft(num - 1)
    out = np.floor(ft(start) + np.arange(num).astype(ft) * delta).astype(np.int64)
    # the last knot is the exact stop, as numpy.linspace pins it: for every R_dt of the shipped configs floor((num-1) * delta)
    # already lands there in both dtypes (tests/test_oracle_cld.py), so this changes nothing there and guards the table's last row elsewhere
    out[-1] = int(stop)
    return out


_R_CACHE = {}


class CLD(object):
    def __init__(self, m_inv=4.0, beta_0=4.0, beta_1=0.0, vv_gamma=0.04, numerical_eps=1e-6, mixed_score=False,
                 used_cache=True, x64=False, is_R_rk=False, R_dt=1e-5):
        self.mixed_score = mixed_score
        self.used_cache = used_cache
        self.x64 = bool(x64)
        self._ft = np.float64 if self.x64 else np.float32  # the dtype jax would compute in (sde_lib.py:112)
        self.m_inv = float(m_inv)
        self.Gamma = 2.0 / np.sqrt(m_inv)
        self.beta_0 = float(beta_0)
        self.beta_1 = float(beta_1)
        self.vv_gamma = float(vv_gamma)
        self.numerical_eps = float(numerical_eps)
        ft = self._ft
        self.R_0 = np.array([[np.sqrt(ft(numerical_eps)), 0.0], [0.0, np.sqrt(ft(vv_gamma / m_inv + numerical_eps))]], dtype=ft)
        self.is_R_rk = bool(is_R_rk)
        self.R_dt = float(R_dt)
        self._xp, self._fp = self._r_knots()
        self.sampling_eps = 1e-3
        self.T = 1.0

    # ---- R(t) ----
    def _r_knots(self):
        key = (self.m_inv, self.beta_0, self.beta_1, self.vv_gamma, self.numerical_eps, self.is_R_rk, self.R_dt, self.x64)
        if self.used_cache and key in _R_CACHE:
            return _R_CACHE[key]
        ft = self._ft
        n = int(1.0 / self.R_dt) + 1  # sde_lib.py:114 (note int(1/1e-5) == 99999)
        table = np.empty((n, 2, 2), dtype=ft)
        scan = _abi.lib().gd_cld_r_table if self.x64 else _abi.lib().gd_cld_r_table_f32
        _abi.check(scan(self.m_inv, self.beta_0, self.beta_1, self.vv_gamma, self.numerical_eps, int(self.is_R_rk), self.R_dt, n,
                        table.ctypes.data_as(ctypes.c_void_p)))
        ts = linspace(0, 1.0 + self.R_dt, n, endpoint=False, ft=ft)  # :116
        idx = linspace_int(0, n - 1, 100_000, ft=ft)  # :117
        _R_CACHE[key] = (ts[idx], table[idx])
        return _R_CACHE[key]

    def v_R(self, t):
        """R at a vector of times, (n, 2, 2): get_interp_fn (sde_lib.py:32-43) in the configured dtype."""
        t = np.atleast_1d(np.asarray(t)).astype(self._ft)
        xp, fp = self._xp, self._fp
        i = np.clip(np.searchsorted(xp, t, side='right'), 1, len(xp) - 1)
        df = fp[i] - fp[i - 1]
        dx = xp[i] - xp[i - 1]
        delta = t - xp[i - 1]
        w = delta / np.where(dx == 0, self._ft(1), dx)
        return np.where((dx == 0)[:, None, None], fp[i], fp[i - 1] + w[:, None, None] * df)

    def s_R(self, t):
        return self.v_R(t)[0]

    def s_invR(self, t):
        return inv_2x2(self.s_R(t))

    def v_invR(self, t):
        return inv_2x2(self.v_R(t))

    def s_cov(self, t):
        R = self.s_R(t)
        return R @ R.T

    # ---- schedule / drift / diffusion ----
    def beta(self, t):
        ft = self._ft
        return ft(self.beta_0) + ft(self.beta_1) * np.asarray(t).astype(ft)

    def beta_int(self, t):
        ft = self._ft
        t = np.asarray(t).astype(ft)
        return ft(self.beta_0) * t + ft(0.5 * self.beta_1) * t ** 2  # :84-85, `0.5 * self.beta_1` is a Python product

    def s_F(self, t):
        ft = self._ft
        b = ft(self.beta(t))
        return np.array([[0.0, b * ft(self.m_inv)], [-b, -ft(self.Gamma) * b * ft(self.m_inv)]], dtype=ft)

    def s_G(self, t):
        ft = self._ft
        b = ft(self.beta(t))
        return np.array([[0.0, 0.0], [0.0, np.sqrt(ft(2 * self.Gamma) * b)]], dtype=ft)

    def vs_psi(self, s_vec, t):
        """Psi(s_n, t) = expm(int_s^t F): e^{-ab/2} [[1+ab/2, a^2 b/4], [-b, 1-ab/2]], a = 2 sqrt(m_inv) (sde_lib.py:182-205)."""
        ft = self._ft
        b = self.beta_int(t) - self.beta_int(s_vec)
        a = ft(2) * np.sqrt(ft(self.m_inv))
        e = np.exp(-a * b / ft(2))
        out = np.empty(np.shape(b) + (2, 2), dtype=ft)
        out[..., 0, 0] = (ft(1) + a * b / ft(2)) * e
        out[..., 0, 1] = (ft(0.25) * a * a * b) * e
        out[..., 1, 0] = -b * e
        out[..., 1, 1] = (ft(1) - a * b / ft(2)) * e
        return out

    def s_psi(self, s, t):
        return self.vs_psi(self._ft(s), t)

    def vv_psi(self, s_vec, t_vec):
        return self.vs_psi(np.asarray(s_vec), np.asarray(t_vec))

    def v_eps_integrand(self, t_vec):
        """0.5 * G @ G @ inv_2x2(R).T at a vector of times (sde_lib.py:207-212): G = diag(0, g), so only row 1 is non-zero and
        its factor is (0.5 g) g, formed left to right."""
        ft = self._ft
        t_vec = np.atleast_1d(np.asarray(t_vec)).astype(ft)
        invT = np.swapaxes(inv_2x2(self.v_R(t_vec)), -1, -2)
        g = np.sqrt(ft(2 * self.Gamma) * self.beta(t_vec))
        out = np.zeros_like(invT)
        out[:, 1, :] = ((ft(0.5) * g) * g)[:, None] * invT[:, 1, :]
        return out

    def s_eps_integrand(self, t):
        return self.v_eps_integrand(t)[0]

    # ---- F = F1 + F2 split of the rotated-frame sampler (sde_lib.py:121-180) ----
    def s_f1_psi(self, s, t):
        """expm(int_s^t F1): a rotation-like matrix, F1 = [[0, beta m_inv], [-beta, 0]]."""
        b = float(self.beta_int(t)) - float(self.beta_int(s))
        sqrt_m, inv_sqrt_m = np.sqrt(1.0 / self.m_inv), np.sqrt(self.m_inv)
        c, sn = np.cos(b * inv_sqrt_m), np.sin(b * inv_sqrt_m)
        return np.array([[c, inv_sqrt_m * sn], [-sqrt_m * sn, c]])

    def s_psi1(self, t):
        return self.s_f1_psi(0.0, t)

    def s_inv_psi1(self, t):
        return self.s_f1_psi(t, 0.0)

    def s_F1(self, t):
        b = float(self.beta(t))
        return np.array([[0.0, b * self.m_inv], [-b, 0.0]])

    def s_F2(self, t):
        b = float(self.beta(t))
        return np.array([[0.0, 0.0], [0.0, -self.Gamma * b * self.m_inv]])

    def mixed_u_matrix(self, t):
        """mixed score: eps_fn adds bmm(invR(t), u with its x part zeroed) = (invR(t) diag(0, 1)) u (models/utils.py:174-176)."""
        m = self.s_invR(t).copy()
        m[:, 0] = 0.0
        return m

    # ---- device-side helpers ----
    def eps2score(self, eps, ts):
        """score = -R(t)^-T eps, per-sample t (sde_lib.py:246-253); eps is a device tensor (B,...,d,2)."""
        from .. import ops
        t = ts.detach().cpu().numpy() if hasattr(ts, 'detach') else np.asarray(ts)
        m = -np.swapaxes(self.v_invR(t), -1, -2)
        return ops.bmm(m.astype(np.float32), eps)

    def prior_sampling(self, rng, shape, device=None):
        """x ~ N(0, I), v ~ N(0, I)/sqrt(m_inv), stacked on the last axis (sde_lib.py:270-274).
        Drawn with torch's generator on the device; jax's threefry stream cannot be reproduced, so parity
        runs pass `u` explicitly."""
        import torch
        device = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        g = torch.Generator(device=device)
        g.manual_seed(int(rng) if not hasattr(rng, 'integers') else int(rng.integers(0, 2 ** 62)))
        xs = torch.randn(tuple(shape), generator=g, device=device, dtype=torch.float32)
        vs = torch.randn(tuple(shape), generator=g, device=device, dtype=torch.float32) / float(np.sqrt(self.m_inv))
        return torch.stack([xs, vs], dim=-1)

    # ---- sampler coefficient tables ----
    def prepare_order0_coef(self, rev_ts, num_item=1000):
        """sde_lib.py:289-306: Psi_i and a 1000-point left Riemann epsilon matrix per step (`integrand * dt` before the sum)."""
        ts = np.asarray(rev_ts).astype(self._ft)
        mean = self.vv_psi(ts[:-1], ts[1:])
        eps = deis.get_order0_eps_coef(self, ts, num_item=num_item)
        return mean, eps

    def prepare_naive_coef(self, rev_ts):
        """sde_lib.py:276-287 (`get_order0_sampler(is_em=True)`): the Euler step of the probability-flow ODE in the
        [mean, eps] form -- Psi_i = I + F(t_i) dt and C_i = 0.5 * G @ G @ inv(R(t_i)).T * dt, dt = t_{i+1} - t_i < 0."""
        ft = self._ft
        ts = np.asarray(rev_ts).astype(ft)
        n = ts.shape[0] - 1
        mean, eps = np.empty((n, 2, 2), dtype=ft), np.empty((n, 2, 2), dtype=ft)
        for i in range(n):
            dt = ts[i + 1] - ts[i]
            mean[i] = np.eye(2, dtype=ft) + self.s_F(ts[i]) * dt
            eps[i] = self.s_eps_integrand(ts[i]) * dt
        return mean, eps

    def get_deis_coef(self, order, rev_timesteps, used_cache=True):
        """(N, order+3, 2, 2) = [Psi_i, C_i0 .. C_i(order+1)] (sde_lib.py:308-319; highest_order = order+1)."""
        ts = np.asarray(rev_timesteps).astype(self._ft)
        x_coef = self.vv_psi(ts[:-1], ts[1:])
        eps_coef = deis.get_ab_eps_coef(self, order + 1, ts, order)
        return np.concatenate([x_coef[:, None], eps_coef], axis=1)


class LSDE(object):
    """Cholesky ("L") parametrisation of the noise prediction (cld_jax/sde_lib.py:469-520): eps_L = L^T R^-T eps_R with
    L = chol(R R^T); the DEIS integrand becomes 1/2 G G L^-T.  Has no s_F, exactly like the reference class."""

    def __init__(self, sde, used_cache=True):
        self.sde = sde
        self.mixed_score = sde.mixed_score
        self.prior_sampling = sde.prior_sampling
        self.v_invR = sde.v_invR
        self.s_G = sde.s_G
        self.used_cache = sde.used_cache and used_cache
        self.x64 = sde.x64
        self.vs_psi = sde.vs_psi
        self.vv_psi = sde.vv_psi
        self.T = sde.T
        self.sampling_eps = sde.sampling_eps

    def v_L(self, t_vec):
        R = self.sde.v_R(t_vec)
        return np.linalg.cholesky(R @ np.swapaxes(R, -1, -2))

    def s_L(self, t):
        return self.v_L(t)[0]

    def epsR2epsL_matrix(self, t):
        """coef of epsR2epsL: L(t)^T inv(R(t)^T)."""
        return self.s_L(t).T @ inv_2x2(self.sde.s_R(t).T)

    def epsR2epsL(self, s_t, eps):
        from . import utils
        return utils.sbmm(self.epsR2epsL_matrix(s_t), eps)

    def v_eps_integrand(self, t_vec):
        t_vec = np.atleast_1d(np.asarray(t_vec, dtype=np.float64))
        invLT = np.swapaxes(inv_2x2(self.v_L(t_vec)), -1, -2)
        out = np.zeros_like(invLT)
        out[:, 1, :] = (self.sde.Gamma * self.sde.beta(t_vec))[:, None] * invLT[:, 1, :]  # 1/2 G G = diag(0, Gamma beta)
        return out

    def s_eps_integrand(self, t):
        return self.v_eps_integrand(t)[0]

    def get_deis_coef(self, order, rev_timesteps, used_cache=True):
        ts = np.asarray(rev_timesteps, dtype=np.float64)
        x_coef = self.vv_psi(ts[:-1], ts[1:])
        eps_coef = deis.get_ab_eps_coef(self, order + 1, ts, order)
        return np.concatenate([x_coef[:, None], eps_coef], axis=1)


_PSI2_CACHE = {}


class MLCLD(object):
    """Rotated-frame CLD (cld_jax/sampling.py:272-320): y = psi1(t)^-1 x removes the oscillatory part F1 of the drift; the
    remaining transition psi2 comes from a 100 000-step RK4 scan (native gd_cld_psi2_table) and linear interpolation."""

    N = 100000

    def __init__(self, sde):
        assert isinstance(sde, CLD)
        assert sde.beta_1 == 0
        self.sde = sde
        self.sampling_eps = sde.sampling_eps
        self.T = sde.T
        self.mixed_score = sde.mixed_score
        self.s_G = sde.s_G
        self.prior_sampling = sde.prior_sampling
        key = (sde.m_inv, sde.beta_0, sde.beta_1)
        if key not in _PSI2_CACHE:
            n = self.N + 1
            table = np.empty((n, 2, 2))
            _abi.check(_abi.lib().gd_cld_psi2_table(sde.m_inv, sde.beta_0, sde.beta_1, 1.0 / self.N, n,
                                                    table.ctypes.data_as(ctypes.c_void_p)))
            _PSI2_CACHE[key] = (np.arange(n) * (1.0 / self.N), table)
        self._xp, self._fp = _PSI2_CACHE[key]

    def v_psi2(self, t):
        t = np.atleast_1d(np.asarray(t, dtype=np.float64))
        xp, fp = self._xp, self._fp
        i = np.clip(np.searchsorted(xp, t, side='right'), 1, len(xp) - 1)
        dx = xp[i] - xp[i - 1]
        w = (t - xp[i - 1]) / dx
        return fp[i - 1] + w[:, None, None] * (fp[i] - fp[i - 1])

    def s_psi2_fn(self, t):
        return self.v_psi2(t)[0]

    def y2x_matrix(self, t):
        return self.sde.s_psi1(t)

    def x2y_matrix(self, t):
        return self.sde.s_inv_psi1(t)

    def y2x(self, y, t):
        from . import utils
        return utils.sbmm(self.y2x_matrix(t), y)

    def x2y(self, x, t):
        from . import utils
        return utils.sbmm(self.x2y_matrix(t), x)

    def vs_psi(self, s_vec, t):
        return self.v_psi2(t) @ inv_2x2(self.v_psi2(s_vec))

    def vv_psi(self, s_vec, t_vec):
        return self.v_psi2(t_vec) @ inv_2x2(self.v_psi2(s_vec))

    def s_psi(self, s, t):
        return self.vs_psi(s, t)[0]

    def v_eps_integrand(self, t_vec):
        t_vec = np.atleast_1d(np.asarray(t_vec, dtype=np.float64))
        sde = self.sde
        b = sde.beta_int(t_vec)
        sqrt_m, inv_sqrt_m = np.sqrt(1.0 / sde.m_inv), np.sqrt(sde.m_inv)
        c, sn = np.cos(b * inv_sqrt_m), np.sin(b * inv_sqrt_m)
        ip1 = np.empty((len(t_vec), 2, 2))  # s_f1_psi(t, 0): the angle changes sign
        ip1[:, 0, 0], ip1[:, 0, 1], ip1[:, 1, 0], ip1[:, 1, 1] = c, -inv_sqrt_m * sn, sqrt_m * sn, c
        invT = np.swapaxes(inv_2x2(sde.v_R(t_vec)), -1, -2)
        gg = np.zeros_like(invT)
        gg[:, 1, :] = (sde.Gamma * sde.beta(t_vec))[:, None] * invT[:, 1, :]  # 1/2 G G^T R^-T
        return ip1 @ gg

    def s_eps_integrand(self, t):
        return self.v_eps_integrand(t)[0]

    def get_deis_coef(self, order, rev_timesteps):
        ts = np.asarray(rev_timesteps, dtype=np.float64)
        x_coef = self.vv_psi(ts[:-1], ts[1:])
        eps_coef = deis.get_ab_eps_coef(self, order + 1, ts, order)
        return np.concatenate([x_coef[:, None], eps_coef], axis=1)


class LambdaSDE(object):
    """The lambda-family of reverse SDEs (cld_jax/sde_lib.py:334-466): lambda = 0 is the probability-flow ODE, 1 the reverse
    diffusion.  \\hat F = F + (1 + lambda^2)/2 G G^T Sigma^-1; the transition \\hat psi and the conditional covariance of a step come
    from the native RK4 scans (gd_cld_lambda_hat_psi_table / gd_cld_lambda_cond_rev_cov), which follow the reference's code
    literally -- including `P @ \\hat F` where its own comment says `P \\hat F^T`, so the "covariance" it feeds to the sampler is
    not symmetric (reference quirks)."""

    def __init__(self, sde, lambda_coef=0.1, use_order0=True, used_cache=True):
        self.sde = sde
        self.mixed_score = sde.mixed_score
        self.prior_sampling = sde.prior_sampling
        self.v_invR = sde.v_invR
        self.used_cache = sde.used_cache and used_cache
        self.x64 = sde.x64
        self.use_order0 = use_order0
        self.lambda_coef = float(lambda_coef)
        self.T, self.sampling_eps = sde.T, sde.sampling_eps
        dt = 1e-5
        n = int(1.0 / dt) + 1
        table = np.empty((n, 2, 2))
        self._rx = np.ascontiguousarray(sde._xp, dtype=np.float64)
        self._rf = np.ascontiguousarray(sde._fp, dtype=np.float64)
        _abi.check(_abi.lib().gd_cld_lambda_hat_psi_table(sde.m_inv, sde.beta_0, sde.beta_1, self.lambda_coef,
                                                          self._rx.ctypes.data_as(ctypes.c_void_p),
                                                          self._rf.ctypes.data_as(ctypes.c_void_p), len(self._rx), dt, n,
                                                          table.ctypes.data_as(ctypes.c_void_p)))
        self._xp, self._fp = np.linspace(0, 1.0 + dt, n, endpoint=False), table

    def v_hat_psi_02t(self, t):
        t = np.atleast_1d(np.asarray(t, dtype=np.float64))
        xp, fp = self._xp, self._fp
        i = np.clip(np.searchsorted(xp, t, side='right'), 1, len(xp) - 1)
        w = (t - xp[i - 1]) / (xp[i] - xp[i - 1])
        return fp[i - 1] + w[:, None, None] * (fp[i] - fp[i - 1])

    def s_hat_F(self, t):
        G = self.sde.s_G(t)
        return self.sde.s_F(t) + 0.5 * (1 + self.lambda_coef ** 2) * G @ G.T @ inv_2x2(self.sde.s_cov(t))

    def s_hat_psi(self, s, t):
        return self.v_hat_psi_02t(t)[0] @ inv_2x2(self.v_hat_psi_02t(s)[0])

    def vs_hat_psi(self, s_vec, t):
        return self.v_hat_psi_02t(t) @ inv_2x2(self.v_hat_psi_02t(s_vec))

    def cond_rev_cov(self, s, t, n_step=10_000):
        out = np.empty((2, 2))
        _abi.check(_abi.lib().gd_cld_lambda_cond_rev_cov(self.sde.m_inv, self.sde.beta_0, self.sde.beta_1, self.lambda_coef,
                                                         self._rx.ctypes.data_as(ctypes.c_void_p),
                                                         self._rf.ctypes.data_as(ctypes.c_void_p), len(self._rx), float(s), float(t),
                                                         int(n_step), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def update_coef(self, s, t):
        x_coef = self.sde.s_psi(s, t)
        eps_coef = (self.s_hat_psi(s, t) - x_coef) @ self.sde.s_R(s)
        return np.stack([x_coef, eps_coef, self.cond_rev_cov(s, t)])

    def get_poly_eps_coef(self, order, rev_timesteps):
        outer = self

        class _sde(object):
            @staticmethod
            def vs_psi(ss, t):
                return outer.vs_hat_psi(ss, t)

            @staticmethod
            def v_eps_integrand(ts):
                ts = np.atleast_1d(np.asarray(ts, dtype=np.float64))
                R = outer.sde.v_R(ts)
                inv_cov = inv_2x2(R @ np.swapaxes(R, -1, -2))
                gg = np.zeros_like(inv_cov)  # 1/2 (1 + lambda^2) G G^T inv_cov: only the v row of G G^T is non-zero
                gg[:, 1, :] = ((1 + outer.lambda_coef ** 2) * outer.sde.Gamma * outer.sde.beta(ts))[:, None] * inv_cov[:, 1, :]
                return gg @ outer.sde.vs_psi(np.zeros_like(ts), ts)

        ab_eps_coef = deis.get_ab_eps_coef(_sde, order + 1, rev_timesteps, order)
        ts = np.asarray(rev_timesteps, dtype=np.float64)[:-1]
        last_term = self.sde.vs_psi(ts, 0.0) @ self.sde.v_R(ts)
        return np.einsum("b...ij,bjk->b...ik", ab_eps_coef, last_term)

    def get_order0_coef(self, rev_timesteps, used_cache=True):
        ts = np.asarray(rev_timesteps, dtype=np.float64)
        return np.stack([self.update_coef(ts[i], ts[i + 1]) for i in range(len(ts) - 1)])

    def get_deis_coef(self, order, rev_timesteps, used_cache=True):
        ts = np.asarray(rev_timesteps, dtype=np.float64)
        if self.use_order0 and order == 0:
            c = self.get_order0_coef(ts)
            return np.stack([c[:, 0], c[:, 1], np.zeros(c[:, 0].shape), c[:, 2]], axis=1)
        x_coef = self.sde.vv_psi(ts[:-1], ts[1:])
        eps_coef = self.get_poly_eps_coef(order, ts)
        covs = np.stack([self.cond_rev_cov(ts[i], ts[i + 1]) for i in range(len(ts) - 1)])
        return np.concatenate([x_coef[:, None], eps_coef, covs[:, None]], axis=1)


def from_config(config):
    return CLD(m_inv=config.model.m_inv, beta_0=config.model.beta_0, beta_1=config.model.beta_1,
               vv_gamma=config.model.vv_gamma, mixed_score=config.model.mixed_score, is_R_rk=config.model.is_R_rk,
               used_cache=config.model.used_cache, R_dt=config.model.R_dt, x64=config.model.x64)
