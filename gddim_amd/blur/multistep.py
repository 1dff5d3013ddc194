"""Scalar-coefficient DEIS (the twin of deis.py) -- blur_jax/multistep.py:4-97.

The reference file is never imported and cannot run against the blur SDE's per-frequency psi
(SURVEY.md section 2.1); it is built here exactly as written, for any `sde` exposing scalar
`psi(t_start_vec, t_end)` and `eps_integrand(t_vec)`:
    c_{i,s} = sum_{n<10000} psi(tau_n, t_{i+1}) g(tau_n) l_{q-s}(tau_n) dtau   (left Riemann, endpoint=False)
and `ab_step` x' = c_0 x + sum_o c_{o+1} eps_o runs as a HIP kernel."""
import numpy as np

from .. import ops
from ..cld.deis import _lagrange_weights

NUM_ITEM = 10000


def get_ab_eps_coef(sde, highest_order, timesteps, order, num_item=NUM_ITEM):
    ts = np.asarray(timesteps, dtype=np.float64)
    N = len(ts) - 1
    out = np.zeros((N, highest_order + 1))
    for i in range(N):
        q = min(i, order)
        t0, t1 = ts[i], ts[i + 1]
        t_inter = t0 + (t1 - t0) * np.arange(num_item) / num_item
        dt = (t1 - t0) / num_item
        base = sde.psi(t_inter, t1) * sde.eps_integrand(t_inter)
        w = _lagrange_weights(t_inter, ts[i - q:i + 1])
        for s in range(q + 1):
            out[i, s] = np.sum(w[q - s] * base) * dt
    return out


def ab_step(x, ei_coef, new_eps, eps_pred):
    return ops.ab_step_scalar(x, ei_coef, new_eps, eps_pred)


# ---- per-frequency completion for the blur SDE ----------------------------------------------------
# In DCT space every frequency (h, w) is a scalar linear SDE y_t = m(t) y_0 + s(t) eps with m = sqrt(alpha) D_t[h, w], s = sqrt(1 -
# alpha): the transition is psi(tau, t) = m(t) / m(tau) and the eps term of the exponential integrator is
#     int_{t_i}^{t_{i+1}} psi(tau, t_{i+1}) kappa(tau) eps(tau) dtau,   psi(tau, t') kappa(tau) = d/dtau [ psi(tau, t') s(tau) ]
# (the identity that makes the order-0 update y' = psi y + (s' - psi s) eps exact for constant eps, blur_jax/sampling.py:69-74).  With
# eps replaced by its Lagrange interpolant through the last q + 1 predictions:
#     c_{i,j} = int d[psi(tau, t_{i+1}) s(tau)] l_j(tau)   ~   sum_n (Phi(tau_{n+1}) - Phi(tau_n)) l_j(tau_n + dtau / 2).
# Exact in Phi, midpoint in l: for q = 0 the sum telescopes to the order-0 coefficient exactly, and sum_j c_{i,j} equals it for any q.
# Same warm start (row i uses order min(i, q)) and slot order (0 = newest) as deis.py / multistep.py above.
def get_ab_eps_coef_freq(sde, highest_order, timesteps, order, num_item=2000):
    """(N, highest_order + 1, 32, 32) eps coefficients and (N, 32, 32) transitions psi_i of the blur SDE."""
    ts = np.asarray(timesteps, dtype=np.float64)
    N = len(ts) - 1
    coef = np.zeros((N, highest_order + 1, 32, 32))
    psi = np.zeros((N, 32, 32))
    for i in range(N):
        q = min(i, order)
        t0, t1 = ts[i], ts[i + 1]
        grid = t0 + (t1 - t0) * np.arange(num_item + 1) / num_item
        m = sde.y_mean_coef(grid)[..., 0]                       # (n + 1, 32, 32)
        phi = m[-1][None] / m * sde.y_std_coef(grid)[:, None, None]
        dphi = phi[1:] - phi[:-1]                               # (n, 32, 32)
        w = _lagrange_weights(0.5 * (grid[1:] + grid[:-1]), ts[i - q:i + 1])  # (q + 1, n)
        for s in range(q + 1):
            coef[i, s] = np.einsum("n,nhw->hw", w[q - s], dphi)
        psi[i] = m[-1] / m[0]
    return coef, psi
