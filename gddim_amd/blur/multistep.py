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
