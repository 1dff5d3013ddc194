"""ORACLE (test infrastructure only): CLD SDE algebra + DEIS coefficients + samplers on CPU.

Restates, in numpy (dtype selectable: float32 like the reference's ``x64=False`` run, or
float64), the following reference code:

  /root/reference/cld_jax/sde_lib.py:17-43    inv_2x2, get_interp_fn
  /root/reference/cld_jax/sde_lib.py:45-331   class CLD (constants, s_F, s_G, s_psi, R scan,
                                              s_eps_integrand, eps2score, prior_sampling,
                                              prepare_order0_coef, get_deis_coef)
  /root/reference/cld_jax/deis.py:19-95       DEIS Adams-Bashforth epsilon coefficients
  /root/reference/cld_jax/deis.py:141-151     multistep_ab_step
  /root/reference/cld_jax/sampling.py:30-39   get_denoising_step
  /root/reference/cld_jax/sampling.py:156-253 get_order0_sampler, _impl_deis_sampler,
                                              get_rev_ts, get_deis_sampler
  /root/reference/cld_jax/utils.py:172-185    get_data_shape, bmm, sbmm

Deliberately literal (python loops per step / per order) rather than fast.
PARITY UNPINNED -- see oracle/__init__.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIB = None


def build_c(force=False):
    """Compile oracle/rscan.c with gcc into oracle/_build/liboracle_rscan.so."""
    os.makedirs(_BUILD, exist_ok=True)
    so = os.path.join(_BUILD, "liboracle_rscan.so")
    src = os.path.join(_HERE, "rscan.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        # -ffp-contract=off: no FMA fusion, keep the float path a plain IEEE restatement.
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_c())
    return _LIB


def inv_2x2(m):
    # sde_lib.py:17-26
    a, b, c, d = m[0, 0], m[0, 1], m[1, 0], m[1, 1]
    coef = 1.0 / (a * d - b * c)
    return np.asarray([[d, -b], [-c, a]], dtype=m.dtype) * m.dtype.type(coef)


def jnp_linspace(start, stop, num, endpoint=True, dtype=np.float64):
    """jnp.linspace of jax 0.2.8 (the reference's pin, Dockerfile:1-15) for a floating dtype, restated from memory of
    jax/_src/numpy/lax_numpy.py: `step = iota / div; out = start * (1 - step) + stop * step` in the computation dtype ("recovers
    the endpoints with float32 arithmetic").  In float32 this differs from numpy's float64-then-round grid by an ulp per point,
    which is 0.6 % of the 1e-5 knot spacing of the R table near t = 1 -- it matters to get_interp_fn's weights."""
    T = np.dtype(dtype).type
    div = (num - 1) if endpoint else num
    if num == 1:
        return np.asarray([start], dtype=T)
    step = np.arange(num, dtype=T) / T(div)
    return T(start) * (T(1) - step) + T(stop) * step


def jnp_linspace_int(start, stop, num, dtype=np.float64):
    """jnp.linspace(start, stop, num, dtype=int): floor(start + iota * delta), delta = (stop - start) / (num - 1), evaluated in
    the floating computation dtype (`dtype`: float32 when x64 is off) and then converted."""
    T = np.dtype(dtype).type
    delta = (T(stop) - T(start)) / T(num - 1)
    return np.floor(T(start) + np.arange(num).astype(T) * delta).astype(np.int64)


def get_interp_fn(xp, fp):
    # sde_lib.py:32-43  (searchsorted side='right', clipped to [1, len-1], linear)
    def _fn(x):
        x = np.asarray(x, dtype=xp.dtype)
        i = np.clip(np.searchsorted(xp, x, side="right"), 1, len(xp) - 1)
        df = fp[i] - fp[i - 1]
        dx = xp[i] - xp[i - 1]
        delta = x - xp[i - 1]
        if np.ndim(x) == 0:
            return fp[i] if dx == 0 else fp[i - 1] + (delta / dx) * df
        w = (delta / np.where(dx == 0, 1.0, dx)).astype(xp.dtype)
        return np.where((dx == 0)[..., None, None], fp[i], fp[i - 1] + w[..., None, None] * df)  # :41 fp[i] on a zero-width cell

    return _fn


class CLD:
    """sde_lib.py:45-331 restated.  ``dtype`` plays the role of the jax x64 switch."""

    def __init__(self, m_inv=4.0, beta_0=4.0, beta_1=0.0, vv_gamma=0.04, numerical_eps=1e-6,
                 mixed_score=False, is_R_rk=False, R_dt=1e-5, dtype=np.float64, sum_order="pairwise"):
        # sum_order: how the 10 000-term `jnp.sum(..., axis=0)` of deis.py:46 / sde_lib.py:296 is accumulated in `dtype`.  XLA's
        # order is not defined by the reference's source: a tree on the GPU backend ("pairwise": numpy's pairwise sum along a
        # contiguous axis), a running loop on the CPU backend ("sequential").  In float32 the two differ by ~2e-6 of the table
        # scale -- as much as float32 differs from float64 (tests/test_oracle_cld.py holds both numbers).
        assert sum_order in ("pairwise", "sequential")
        self.sum_order = sum_order
        self.dtype = np.dtype(dtype)
        T = self.dtype.type
        self.mixed_score = mixed_score
        self.m_inv = T(m_inv)
        self.Gamma = T(2.0 / np.sqrt(m_inv))  # :57
        self.beta_0 = T(beta_0)
        self.beta_1 = T(beta_1)
        # :61-66: jnp.sqrt of Python floats -- `vv_gamma / self.m_inv + numerical_eps` is a Python (double) expression
        self.R_0 = np.asarray([[np.sqrt(T(numerical_eps)), 0],
                               [0, np.sqrt(T(vv_gamma / m_inv + numerical_eps))]], dtype=self.dtype)
        self.is_R_rk = bool(is_R_rk)
        self.R_dt = float(R_dt)
        self.sampling_eps = 1e-3  # :76
        self.T = 1.0  # :77
        self.s_R = self._get_s_R_fn(self.is_R_rk, self.R_dt)

    # ---- schedule ----
    def _beta(self, t):
        return self.beta_0 + self.beta_1 * t  # :80-81

    def _beta_int(self, t):
        return self.beta_0 * t + 0.5 * self.beta_1 * t ** 2  # :84-85

    # ---- R(t) ----
    def _get_s_R_fn(self, is_rk, R_dt):
        # :93-118.  n = int(1/R_dt) + 1 grid points on linspace(0, 1 + R_dt, n, endpoint=False).
        n = int(1.0 / R_dt) + 1
        ts = jnp_linspace(0, 1.0 + R_dt, n, endpoint=False, dtype=self.dtype)
        out = np.empty((n, 2, 2), dtype=self.dtype)
        lib = _lib()
        if self.dtype == np.float32:
            fn, cT = lib.oracle_rscan_f32, ctypes.c_float
        else:
            fn, cT = lib.oracle_rscan_f64, ctypes.c_double
        fn.restype = None
        fn.argtypes = [ctypes.c_void_p, cT, cT, cT, cT, cT, ctypes.c_void_p, ctypes.c_size_t,
                       ctypes.c_int, ctypes.c_void_p]
        R0 = np.ascontiguousarray(self.R_0)
        fn(R0.ctypes.data, cT(self.m_inv), cT(self.Gamma), cT(self.beta_0), cT(self.beta_1),
           cT(R_dt), ts.ctypes.data, n, int(is_rk), out.ctypes.data)
        self.R_table_ts, self.R_table = ts, out
        idx = jnp_linspace_int(0, n - 1, 100_000, dtype=self.dtype)  # :117
        return get_interp_fn(ts[idx], out[idx])

    def v_R(self, ts):
        return self.s_R(np.asarray(ts, dtype=self.dtype))

    # ---- drift / diffusion / transition ----
    def s_F(self, t):
        b = self._beta(self.dtype.type(t))
        return np.asarray([[0.0, b * self.m_inv], [-b, -self.Gamma * b * self.m_inv]], dtype=self.dtype)  # :214-224

    def s_G(self, t):
        b = self._beta(self.dtype.type(t))
        return np.asarray([[0.0, 0.0], [0.0, np.sqrt(2 * self.Gamma * b)]], dtype=self.dtype)  # :226-234

    def s_psi(self, s, t):
        # :182-205  expm(int_s^t F) in closed form
        T = self.dtype.type
        beta_int = self._beta_int(T(t)) - self._beta_int(T(s))
        a = 2 * np.sqrt(self.m_inv)
        tt = beta_int
        coef = np.exp(-a * tt / 2)
        return np.asarray([[1 + a * tt / 2, 0.25 * a * a * tt], [-tt, 1 - a * tt / 2]], dtype=self.dtype) * coef

    def s_invR(self, t):
        return inv_2x2(self.s_R(self.dtype.type(t)))  # :236-239

    def s_eps_integrand(self, s_t):
        G = self.s_G(s_t)
        return (0.5 * G @ G @ inv_2x2(self.s_R(self.dtype.type(s_t))).T).astype(self.dtype)  # :207-212

    def eps2score(self, eps, ts):
        # :246-253  score = -R^{-T} eps, per-sample t
        out = np.empty_like(eps)
        for b in range(eps.shape[0]):
            inv_r = inv_2x2(self.s_R(self.dtype.type(ts[b])))
            out[b] = np.einsum("ji,...dj->...di", -inv_r, eps[b])
        return out

    def prior_sampling(self, rng, shape):
        # :270-274 (jax threefry cannot be reproduced: numpy Generator stands in)
        xs = rng.standard_normal(shape).astype(self.dtype)
        vs = rng.standard_normal(shape).astype(self.dtype) / np.sqrt(self.m_inv)
        return np.stack([xs, vs], axis=-1)

    # ---- order-0 coefficients (sde_lib.py:289-306) ----
    def prepare_order0_coef(self, rev_ts, num_item=1000):
        rev_ts = np.asarray(rev_ts, dtype=self.dtype)
        N = len(rev_ts) - 1
        mean = np.stack([self.s_psi(rev_ts[i], rev_ts[i + 1]) for i in range(N)])
        eps = np.empty((N, 2, 2), dtype=self.dtype)
        for i in range(N):
            cur_t, next_t = rev_ts[i], rev_ts[i + 1]
            dt = (next_t - cur_t) / num_item
            t_inter = jnp_linspace(cur_t, next_t, num_item, endpoint=False, dtype=self.dtype)
            psi_coef = vs_psi(self, t_inter, next_t)
            integrand = v_eps_integrand(self, t_inter)
            eps[i] = reduce0(np.einsum("bij,bjk->bik", psi_coef, integrand * dt).astype(self.dtype), self.sum_order)
        return mean, eps

    # ---- DEIS coefficients (sde_lib.py:308-319 + deis.py:19-95) ----
    def prepare_naive_coef(self, rev_ts):
        # sde_lib.py:276-287
        rev_ts = np.asarray(rev_ts, dtype=self.dtype)
        mean = np.stack([np.eye(2, dtype=self.dtype) + self.s_F(c) * (n - c) for c, n in zip(rev_ts[:-1], rev_ts[1:])])
        eps = np.stack([self.dtype.type(0.5) * self.s_G(c) @ self.s_G(c) @ inv_2x2(self.s_R(c)).T * (n - c)
                        for c, n in zip(rev_ts[:-1], rev_ts[1:])])
        return mean, eps

    def get_deis_coef(self, order, rev_timesteps):
        rev_timesteps = np.asarray(rev_timesteps, dtype=self.dtype)
        x_coef = np.stack([self.s_psi(rev_timesteps[i], rev_timesteps[i + 1])
                           for i in range(len(rev_timesteps) - 1)])
        eps_coef = get_ab_eps_coef(self, order + 1, rev_timesteps, order)
        return np.concatenate([x_coef[:, None], eps_coef], axis=1)


def reduce0(a, order):
    """jnp.sum(a, axis=0) in a's dtype with a stated accumulation order (see CLD.__init__)."""
    if order == "sequential":
        acc = np.zeros(a.shape[1:], dtype=a.dtype)
        for row in a:  # a running sum, one row at a time
            acc = acc + row
        return acc
    return np.sum(np.ascontiguousarray(np.moveaxis(a, 0, -1)), axis=-1)


def single_poly_coef(t_val, ts_poly, coef_idx):
    # deis.py:29-36  Lagrange basis l_j(tau) on nodes ts_poly
    num = t_val - ts_poly
    denum = ts_poly[coef_idx] - ts_poly
    num = num.copy()
    denum = denum.copy()
    num[coef_idx] = 1.0
    denum[coef_idx] = 1.0
    return np.prod(num) / np.prod(denum)


def _eps_single_coef(sde, t_start, t_end, ts_poly, coef_idx, num_item=10000):
    # deis.py:19-27 + 41-47: left Riemann sum, num_item points, endpoint=False
    T = sde.dtype.type
    dt = (t_end - t_start) / T(num_item)
    t_inter = jnp_linspace(t_start, t_end, num_item, endpoint=False, dtype=sde.dtype)
    # sde.vs_psi(t_inter, t_end) / sde.v_eps_integrand(t_inter): the LSDE / MLCLD restatements (oracle/cld_ext.py) bring
    # their own, the plain CLD uses the closed forms below
    psi_coef = sde.vs_psi(t_inter, t_end) if hasattr(sde, "vs_psi") else vs_psi(sde, t_inter, t_end)                    # (n,2,2)
    integrand = sde.v_eps_integrand(t_inter) if hasattr(sde, "v_eps_integrand") else v_eps_integrand(sde, t_inter)      # (n,2,2)
    prod = np.einsum("bij,bjk->bik", psi_coef, integrand)
    poly = vec_poly_coef(t_inter, ts_poly, coef_idx)   # (n,)
    acc = reduce0((prod * poly[:, None, None]).astype(sde.dtype), getattr(sde, "sum_order", "pairwise"))  # XLA's order: unknowable
    return (acc * dt).astype(sde.dtype)


def vs_psi(sde, s_vec, t):
    """vmap(s_psi, in_axes=(0, None)) -- sde_lib.py:74, elementwise form of :182-205."""
    T = sde.dtype.type
    beta_int = sde._beta_int(T(t)) - sde._beta_int(s_vec)
    a = 2 * np.sqrt(sde.m_inv)
    coef = np.exp(-a * beta_int / 2)
    out = np.empty(s_vec.shape + (2, 2), dtype=sde.dtype)
    out[..., 0, 0] = (1 + a * beta_int / 2) * coef
    out[..., 0, 1] = (0.25 * a * a * beta_int) * coef
    out[..., 1, 0] = (-beta_int) * coef
    out[..., 1, 1] = (1 - a * beta_int / 2) * coef
    return out


def v_eps_integrand(sde, t_vec):
    """vmap(s_eps_integrand) -- sde_lib.py:75,207-212: 0.5 G G inv(R)^T, G = diag(0, g)."""
    Rs = sde.v_R(t_vec)
    det = Rs[:, 0, 0] * Rs[:, 1, 1] - Rs[:, 0, 1] * Rs[:, 1, 0]
    coef = 1.0 / det
    inv = np.empty_like(Rs)
    inv[:, 0, 0] = Rs[:, 1, 1] * coef
    inv[:, 0, 1] = -Rs[:, 0, 1] * coef
    inv[:, 1, 0] = -Rs[:, 1, 0] * coef
    inv[:, 1, 1] = Rs[:, 0, 0] * coef
    invT = np.swapaxes(inv, 1, 2)
    g2 = (2 * sde.Gamma * sde._beta(t_vec)).astype(sde.dtype)  # (G G)[1,1]
    out = np.zeros_like(Rs)
    out[:, 1, :] = 0.5 * g2[:, None] * invT[:, 1, :]
    return out.astype(sde.dtype)


def vec_poly_coef(t_vals, ts_poly, coef_idx):
    """vmap(single_poly_coef, (0, None, None)) -- deis.py:38."""
    num = t_vals[:, None] - ts_poly[None, :]
    denum = ts_poly[coef_idx] - ts_poly
    num[:, coef_idx] = 1.0
    denum = denum.copy()
    denum[coef_idx] = 1.0
    return np.prod(num, axis=1) / np.prod(denum)


def _eps_coef(sde, highest_order, order, t_start, t_end, ts_poly, num_item=10000):
    # deis.py:49-59: slot s uses coef_idx = flip(arange(order+1))[s]  (slot 0 = newest node)
    rtn = np.zeros((highest_order + 1, 2, 2), dtype=sde.dtype)
    ts_poly = ts_poly[:order + 1]
    for s, j in enumerate(np.flip(np.arange(order + 1))):
        rtn[s] = _eps_single_coef(sde, t_start, t_end, ts_poly, int(j), num_item)
    return rtn


def get_ab_eps_coef(sde, highest_order, timesteps, order, num_item=10000):
    # deis.py:61-95 (recursive warm start)
    timesteps = np.asarray(timesteps, dtype=sde.dtype)
    if order == 0:
        rows = []
        for i in range(len(timesteps) - 1):
            rows.append(_eps_coef(sde, highest_order, 0, timesteps[i], timesteps[i + 1],
                                  timesteps[i:i + 1], num_item))
        return np.stack(rows)
    prev_coef = get_ab_eps_coef(sde, highest_order, timesteps[:order + 1], order - 1, num_item)
    rows = []
    for j in range(len(timesteps) - order - 1):
        ts_poly = timesteps[j:j + order + 1]
        rows.append(_eps_coef(sde, highest_order, order, timesteps[order + j], timesteps[order + j + 1],
                              ts_poly, num_item))
    if rows:
        return np.concatenate([prev_coef, np.stack(rows)], axis=0)
    return prev_coef


def multistep_ab_step(x, deis_coef, new_eps, eps_pred):
    # deis.py:141-151
    x_coef, eps_coef = deis_coef[0], deis_coef[1:]
    full_eps = np.concatenate([new_eps[None], eps_pred])
    linear_term = np.einsum("ij,b...j->b...i", x_coef, x)
    eps_term = np.einsum("oij,o...j->...i", eps_coef, full_eps)
    return linear_term + eps_term, full_eps[:-1]


def sbmm(b1, b2):
    return np.einsum("ij,b...dj->b...di", b1, b2)  # utils.py:183-185


def bmm(b1, b2):
    return np.einsum("bij,b...dj->b...di", b1, b2)  # utils.py:179-181


def get_rev_ts(sde, ts_order, num_step, dtype=None):
    # sampling.py:241-249
    dtype = np.dtype(dtype or sde.dtype)
    T = dtype.type
    lin = jnp_linspace(np.power(T(sde.T), T(1.0 / ts_order)), np.power(T(sde.sampling_eps), T(1.0 / ts_order)),
                       num_step + 1, dtype=dtype)
    if isinstance(ts_order, (int, np.integer)) and ts_order == 2:
        return lin * lin  # jnp.power with a Python int exponent is lax.integer_pow (a multiplication chain)
    return np.power(lin, T(ts_order)).astype(dtype)


def get_denoising_step(sde, eps_fn, t, denoising_eps):
    # sampling.py:30-39
    cur_F, cur_G = sde.s_F(t), sde.s_G(t)

    def step_fn(u):
        ones = np.ones(u.shape[0], dtype=u.dtype)
        dt = denoising_eps * (-1)
        cur_eps = eps_fn(u, ones * t)
        cur_score = sde.eps2score(cur_eps, ones * t)
        return u + sbmm(cur_F, u) * dt - sbmm(cur_G @ cur_G, cur_score) * dt

    return step_fn


def deis_sampler(sde, eps_fn, u, nfe, deis_order, ts_order=2, denoising=False, inverse_scaler=None,
                 deis_coef=None, rev_ts=None, trace=None):
    """sampling.py:204-253 (`sampler`, non-pmapped).  eps_fn(u, t_vec) -> eps, numpy in/out.
    `u` is the initial noise (the parity hook); returns (x, v, nfe)."""
    num_step = nfe - 1 if denoising else nfe
    if rev_ts is None:
        rev_ts = get_rev_ts(sde, ts_order, num_step)
    if deis_coef is None:
        deis_coef = sde.get_deis_coef(deis_order, rev_ts)
    ones = np.ones(u.shape[0], dtype=u.dtype)
    eps_pred = np.stack([u] * (deis_order + 1))  # :225
    for i in range(num_step):  # :226 fori_loop
        cur_eps = eps_fn(u, ones * u.dtype.type(rev_ts[i]))
        u, eps_pred = multistep_ab_step(u, deis_coef[i].astype(u.dtype), cur_eps, eps_pred)
        if trace is not None:
            trace.append(u.copy())
    if denoising:
        u = get_denoising_step(sde, eps_fn, sde.sampling_eps, sde.sampling_eps)(u)
    x, v = u[..., 0], u[..., 1]
    if inverse_scaler is not None:
        x = inverse_scaler(x)
    return x, v, nfe


def order0_sampler(sde, eps_fn, u, nfe, denoising=False, inverse_scaler=None, coefs=None, is_em=False):
    """sampling.py:156-202; ts_order is hard-coded to 2 there (:162); is_em selects prepare_naive_coef (:171-174)."""
    num_step = nfe - 1 if denoising else nfe
    rev_ts = get_rev_ts(sde, 2, num_step)
    if coefs is not None:
        mean_matrix, eps_matrix = coefs
    else:
        mean_matrix, eps_matrix = sde.prepare_naive_coef(rev_ts) if is_em else sde.prepare_order0_coef(rev_ts)
    ones = np.ones(u.shape[0], dtype=u.dtype)
    for i in range(num_step):
        u = sbmm(mean_matrix[i].astype(u.dtype), u) + sbmm(eps_matrix[i].astype(u.dtype),
                                                            eps_fn(u, ones * u.dtype.type(rev_ts[i])))
    if denoising:
        u = get_denoising_step(sde, eps_fn, sde.sampling_eps, sde.sampling_eps)(u)
    x, v = u[..., 0], u[..., 1]
    if inverse_scaler is not None:
        x = inverse_scaler(x)
    return x, v, nfe


def get_data_shape(config):
    # utils.py:172-177
    if "ps" in config.data.dataset:
        return (config.data.dim,)
    return (config.data.image_size, config.data.image_size, config.data.num_channels)


def inverse_scaler_centered(x):
    return (x + 1.0) / 2.0  # datasets.py:34-40 with data.centered=True
