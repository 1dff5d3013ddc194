"""ORACLE (test infrastructure only): blurring-diffusion gDDIM on CPU (numpy).

Restates:
  /root/reference/blur_jax/sde_lib.py:18-51,79-97,134-154   class SDE (sampling members)
  /root/reference/blur_jax/sde_lib.py:53-77                 psi / G / eps_integrand (used only by the
                                                            scalar multistep check, with D_t == 1)
  /root/reference/blur_jax/blur.py:11-106                   orthonormal DCT-II / inverse via FFT
  /root/reference/blur_jax/sampling.py:42-89                get_rev_ts, get_order0_sampler
  /root/reference/blur_jax/multistep.py:4-97                scalar DEIS coefficients + ab_step
  /root/reference/blur_jax/models/utils.py:155-160          get_yeps_fn

The vendored blur_jax/fft.py is jax.numpy.fft (lax.fft); numpy.fft has the same definition.
PARITY UNPINNED -- see oracle/__init__.py.
"""
import numpy as np


# ---- DCT through a complex FFT, exactly the construction of blur.py ----
def _impl_dct(x):
    # blur.py:11-37 (1-D, along axis 0 of a vector)
    N = len(x)
    v0 = x[::2]
    v1 = x[1::2][::-1]
    v = np.concatenate([v0, v1])
    fft_value = np.fft.fft(v)
    k = -np.arange(N, dtype=x.dtype) * np.pi / (2 * N)
    W_r, W_i = np.cos(k), np.sin(k)
    V = fft_value.real * W_r - fft_value.imag * W_i
    factor = np.concatenate([np.full((1,), np.sqrt(N) * 2, x.dtype), np.full((N - 1,), np.sqrt(N * 2), x.dtype)])
    return (2 * V / factor).astype(x.dtype)


def _impl_idct(X):
    # blur.py:51-85
    N = X.shape[-1]
    X_v = X / 2
    factor = np.concatenate([np.full((1,), np.sqrt(N) * 2, X.dtype), np.full((N - 1,), np.sqrt(N * 2), X.dtype)])
    X_v = X_v * factor
    k = np.arange(N, dtype=X.dtype) * np.pi / (2 * N)
    W_r, W_i = np.cos(k), np.sin(k)
    V_t_r = X_v
    V_t_i = np.concatenate([X_v[:1] * 0, -X_v[::-1][:-1]])
    V_r = V_t_r * W_r - V_t_i * W_i
    V_i = V_t_r * W_i + V_t_i * W_r
    v = np.fft.irfft(V_r + 1j * V_i, n=N)
    value = np.stack([v[:N - (N // 2)], v[::-1][:N // 2]]).T
    return value.flatten().astype(X.dtype)


def _img_dct(img):
    # blur.py:41-44: rows (W axis) first, then columns
    w_img = np.stack([_impl_dct(r) for r in img])
    h_img = np.stack([_impl_dct(r) for r in w_img.T])
    return h_img.T


def _img_idct(img):
    w_img = np.stack([_impl_idct(r) for r in img])
    h_img = np.stack([_impl_idct(r) for r in w_img.T])
    return h_img.T


def batch_img_dct(xs):
    # blur.py:99-102  NHWC -> per (b, c) plane
    xs = np.asarray(xs)
    out = np.empty_like(xs)
    for b in range(xs.shape[0]):
        for c in range(xs.shape[3]):
            out[b, :, :, c] = _img_dct(xs[b, :, :, c])
    return out


def batch_img_idct(ys):
    ys = np.asarray(ys)
    out = np.empty_like(ys)
    for b in range(ys.shape[0]):
        for c in range(ys.shape[3]):
            out[b, :, :, c] = _img_idct(ys[b, :, :, c])
    return out


class SDE:
    """blur_jax/sde_lib.py:18-154 (sampling members)."""

    def __init__(self, min_scale=0.001, sigma_blur_max=10.0, sampling_eps=1e-5, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        self.min_scale = min_scale
        self.sigma_blur_max = sigma_blur_max
        self.sampling_eps = sampling_eps
        img_dim = 32  # :24 (hard-wired)
        freqs = (np.pi * np.linspace(0, img_dim - 1, img_dim) / img_dim).astype(self.dtype)
        self.labda = freqs[None, :, None, None] ** 2 + freqs[None, None, :, None] ** 2
        self.alpha_start = self.t2alpha_fn(self.dtype.type(0.0))

    @property
    def T(self):
        return 1.0

    @property
    def sampling_T(self):
        return self.rho2t(80.0)

    def t2alpha_fn(self, t):
        return np.cos((t + 0.004) / 1.008 * np.pi / 2) ** 2  # :37-40

    def alpha2t_fn(self, alpha):
        return np.arccos(np.sqrt(alpha)) * 2 / np.pi * 1.008 - 0.004  # :42-45

    def rho2t(self, rho):
        num = self.alpha_start
        denum = (rho + np.sqrt(1 - self.alpha_start)) ** 2 + self.alpha_start
        return self.alpha2t_fn(num / denum)  # :47-51

    def get_frequency_scaling(self, t):
        # :79-88, t: (B,)
        t = np.asarray(t, dtype=self.dtype)
        sigma_blur = self.sigma_blur_max * np.sin(t * np.pi / 2) ** 2
        dissipation_time = sigma_blur ** 2 / 2
        logits = dissipation_time[:, None, None, None] * self.labda
        scaling = np.exp(-logits) * (1 - self.min_scale)
        return (scaling + self.min_scale).astype(self.dtype)

    def y_mean_coef(self, ts):
        ts = np.asarray(ts, dtype=self.dtype)
        return np.sqrt(self.t2alpha_fn(ts))[:, None, None, None] * self.get_frequency_scaling(ts)  # :90-93

    def y_std_coef(self, ts):
        ts = np.asarray(ts, dtype=self.dtype)
        return np.sqrt(1 - self.t2alpha_fn(ts))  # :95-97

    def prior_sampling(self, rng, shape):
        return rng.standard_normal(shape).astype(self.dtype)  # :134-135

    def x2y(self, xs):
        return batch_img_dct(xs)

    def y2x(self, ys):
        return batch_img_idct(ys)

    def encode_t(self, t):
        return 999 * t


def get_rev_ts(sde, ts_order, num_step):
    # blur_jax/sampling.py:42-51
    return np.power(np.linspace(np.power(sde.sampling_T, 1.0 / ts_order), np.power(sde.sampling_eps, 1.0 / ts_order),
                                num_step + 1), ts_order).astype(sde.dtype)


def order0_sampler(sde, x_eps_fn, y, nfe, ts_order=2, inverse_scaler=None, trace=None):
    """blur_jax/sampling.py:53-80 (`sampler`); x_eps_fn(x, t_vec) is the U-Net eps in pixel space."""
    rev_ts = get_rev_ts(sde, ts_order, nfe)
    B = y.shape[0]
    ones = np.ones(B, dtype=y.dtype)

    def y_eps_fn(yc, t):  # models/utils.py:155-160
        return sde.x2y(x_eps_fn(sde.y2x(yc), t))

    def bm(a, b):  # utils.batch_mul
        return a.reshape((B,) + (1,) * (b.ndim - 1)) * b if a.ndim == 1 else a * b

    for i in range(nfe):
        cur_t, next_t = y.dtype.type(rev_ts[i]), y.dtype.type(rev_ts[i + 1])
        y_eps = y_eps_fn(y, cur_t * ones)
        y_mean_coef, y_std_coef = sde.y_mean_coef(cur_t * ones).astype(y.dtype), sde.y_std_coef(cur_t * ones).astype(y.dtype)
        y_0 = 1.0 / y_mean_coef * (y - bm(y_std_coef, y_eps))
        y = sde.y_mean_coef(next_t * ones).astype(y.dtype) * y_0 + bm(sde.y_std_coef(next_t * ones).astype(y.dtype), y_eps)
        if trace is not None:
            trace.append(y.copy())
    x0 = sde.y2x(y)
    x = inverse_scaler(x0) if inverse_scaler is not None else x0
    return x, nfe


# ---- scalar DEIS (blur_jax/multistep.py), duck-typed sde.psi(t_start_vec, t_end) / sde.eps_integrand(t_vec) ----
class ScalarVPSDE:
    """The cosine-alpha VP part of the blur SDE with D_t == 1, as scalar psi / eps_integrand.

    From sde_lib.py:53-77 with get_frequency_scaling == 1:
      psi(s, t) = sqrt(alpha(t) / alpha(s));  G^2 = dalpha/dt * (-1 + (1 - 1/alpha)) = -dalpha/dt / alpha
      eps_integrand = 0.5 G^2 / sqrt(1 - alpha)
    (`dalpha_dt_fn` is referenced at :60 but never defined in the reference; the analytic
    derivative of t2alpha_fn is used.)"""

    def __init__(self, dtype=np.float64):
        self.dtype = np.dtype(dtype)

    def alpha(self, t):
        return np.cos((t + 0.004) / 1.008 * np.pi / 2) ** 2

    def dalpha_dt(self, t):
        a = (t + 0.004) / 1.008 * np.pi / 2
        return -2 * np.cos(a) * np.sin(a) * (np.pi / 2 / 1.008)

    def psi(self, t_start, t_end):
        return np.sqrt(self.alpha(t_end) / self.alpha(t_start))

    def eps_integrand(self, t):
        g2 = -self.dalpha_dt(t) / self.alpha(t)
        return 0.5 * g2 / np.sqrt(1 - self.alpha(t))


def ms_single_poly_coef(t_val, ts_poly, coef_idx):
    num = t_val - ts_poly  # multistep.py:16-27
    denum = ts_poly[coef_idx] - ts_poly
    num = num.copy()
    denum = denum.copy()
    num[coef_idx] = 1.0
    denum[coef_idx] = 1.0
    return np.prod(num) / np.prod(denum)


def ms_one_coef(sde, t_start, t_end, ts_poly, coef_idx, num_item=10000):
    # multistep.py:4-13,32-42
    dt = (t_end - t_start) / num_item
    t_inter = np.linspace(t_start, t_end, num_item, endpoint=False)
    integrand = sde.psi(t_inter, t_end) * sde.eps_integrand(t_inter)
    poly = np.array([ms_single_poly_coef(tv, ts_poly, coef_idx) for tv in t_inter])
    return np.sum(integrand * poly) * dt


def ms_coef_per_step(sde, highest_order, order, t_start, t_end, ts_poly, num_item=10000):
    rtn = np.zeros((highest_order + 1,))  # multistep.py:44-56
    ts_poly = ts_poly[:order + 1]
    for s, j in enumerate(np.flip(np.arange(order + 1))):
        rtn[s] = ms_one_coef(sde, t_start, t_end, ts_poly, int(j), num_item)
    return rtn


def ms_get_ab_eps_coef(sde, highest_order, timesteps, order, num_item=10000):
    timesteps = np.asarray(timesteps)  # multistep.py:58-92
    if order == 0:
        return np.stack([ms_coef_per_step(sde, highest_order, 0, timesteps[i], timesteps[i + 1], timesteps[i:i + 1], num_item)
                         for i in range(len(timesteps) - 1)])
    prev_coef = ms_get_ab_eps_coef(sde, highest_order, timesteps[:order + 1], order - 1, num_item)
    rows = [ms_coef_per_step(sde, highest_order, order, timesteps[order + j], timesteps[order + j + 1],
                             timesteps[j:j + order + 1], num_item)
            for j in range(len(timesteps) - order - 1)]
    return np.concatenate([prev_coef, np.stack(rows)], axis=0) if rows else prev_coef


def ms_ab_step(x, ei_coef, new_eps, eps_pred):
    # multistep.py:94-97
    x_coef, eps_coef = ei_coef[0], ei_coef[1:]
    full_eps = np.concatenate([new_eps[None], eps_pred])
    eps_term = np.einsum("i,i...->...", eps_coef, full_eps)
    return x_coef * x + eps_term, full_eps[:-1]


# ---- per-frequency multistep sampler: NOT a restatement of reference code --------------------------------------------------------
# The reference never runs blur_jax/multistep.py and ships no multistep blur sampler.  This is an independent
# float64 formulation of the product's extension, used to check it: every DCT frequency is a scalar linear SDE with transition
# psi(tau, t) = m(t)/m(tau) (m = y_mean_coef) and noise level s = y_std_coef, whose eps-term integrand is d/dtau [psi(tau, t') s(tau)].
def freq_deis_coef(sde, timesteps, order, quad=None):
    """(N, order + 1, 32, 32) coefficients (slot 0 = newest eps) by adaptive quadrature of l_j(tau) dPhi/dtau (scipy), warm start
    with order min(i, order) -- independent of the product's midpoint rule."""
    import scipy.integrate
    ts = np.asarray(timesteps, dtype=np.float64)
    N = len(ts) - 1
    out = np.zeros((N, order + 1, 32, 32))
    for i in range(N):
        q = min(i, order)
        nodes = ts[i - q:i + 1]
        t0, t1 = ts[i], ts[i + 1]
        m1 = sde.y_mean_coef(np.array([t1]))[0, ..., 0]

        def dphi(tau, h=1e-6):
            def phi(t):
                return m1 / sde.y_mean_coef(np.array([t]))[0, ..., 0] * sde.y_std_coef(np.array([t]))[0]
            return (phi(tau + h) - phi(tau - h)) / (2 * h)

        for s in range(q + 1):
            j = q - s

            def integrand(tau):
                l = np.prod([(tau - nodes[k]) / (nodes[j] - nodes[k]) for k in range(q + 1) if k != j]) if q else 1.0
                return (dphi(tau) * l).ravel()

            val, _ = scipy.integrate.quad_vec(integrand, t0, t1, epsabs=1e-10, epsrel=1e-9)
            out[i, s] = val.reshape(32, 32)
    return out


def freq_deis_sampler(sde, x_eps_fn, y, nfe, order, ts_order=2, inverse_scaler=None, coef=None, psi=None):
    """y' = psi_i y + sum_o c_{i,o} eps_o per frequency; eps history starts from the state (like the reference's deis loops)."""
    rev_ts = get_rev_ts(sde, ts_order, nfe)
    B = y.shape[0]
    ones = np.ones(B, dtype=y.dtype)
    m = sde.y_mean_coef(rev_ts)[..., 0]
    if psi is None:
        psi = m[1:] / m[:-1]
    hist = [y.copy() for _ in range(order)]
    for i in range(nfe):
        e = sde.x2y(x_eps_fn(sde.y2x(y), y.dtype.type(rev_ts[i]) * ones))
        full = [e] + hist
        y = psi[i][None, :, :, None] * y + sum(coef[i, o][None, :, :, None] * full[o] for o in range(order + 1))
        hist = full[:-1]
    x0 = sde.y2x(y)
    return (inverse_scaler(x0) if inverse_scaler is not None else x0), nfe
