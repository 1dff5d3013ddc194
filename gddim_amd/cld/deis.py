"""DEIS Adams-Bashforth coefficients (host, float64) and the fused AB step (HIP).

Replaces cld_jax/deis.py:19-95 (coefficient generator) and :141-151 (multistep_ab_step).

The coefficient of step i (order q, nodes t_{i-q..i}) and slot s is the reference's own quadrature,
    C_{i,s} = sum_{n<10000} Psi(tau_n, t_{i+1}) . 1/2 G G R(tau_n)^-T . l_{q-s}(tau_n) . dtau,
a LEFT Riemann sum on linspace(t_i, t_{i+1}, 10000, endpoint=False) -- kept on purpose: its O(1/n) error is
part of the reference's results.  Rows i < q use order i (warm start), and the table is
zero-padded to highest_order+1 slots.  Unlike the reference, the integrand Psi . g(tau) is evaluated once per
step and shared by the q+1 Lagrange weights, and everything is float64 (the reference is float32).
"""
import numpy as np

from .. import ops

NUM_ITEM = 10000


def _lagrange_weights(t_inter, nodes):
    """(len(nodes), n): l_j(tau) for every node j."""
    q1 = len(nodes)
    w = np.ones((q1, t_inter.shape[0]))
    for j in range(q1):
        for k in range(q1):
            if k != j:
                w[j] *= (t_inter - nodes[k]) / (nodes[j] - nodes[k])
    return w


def get_ab_eps_coef(sde, highest_order, timesteps, order, num_item=NUM_ITEM):
    """(N, highest_order+1, 2, 2); slot 0 multiplies the newest epsilon."""
    ts = np.asarray(timesteps, dtype=np.float64)
    N = len(ts) - 1
    out = np.zeros((N, highest_order + 1, 2, 2))
    for i in range(N):
        q = min(i, order)  # warm start: the recursion of deis.py:71-95 gives row i order min(i, order)
        t0, t1 = ts[i], ts[i + 1]
        t_inter = t0 + (t1 - t0) * np.arange(num_item) / num_item
        dt = (t1 - t0) / num_item
        base = np.einsum("nij,njk->nik", sde.vs_psi(t_inter, t1), sde.v_eps_integrand(t_inter))
        w = _lagrange_weights(t_inter, ts[i - q:i + 1])  # node index j <-> t_{i-q+j}
        for s in range(q + 1):
            out[i, s] = np.einsum("n,nij->ij", w[q - s], base) * dt
    return out


def multistep_ab_step(x, deis_coef, new_eps, eps_pred):
    """x' = Psi x + sum_o C_o full_eps[o], full_eps = [new_eps, *eps_pred]; returns (x', full_eps[:-1])."""
    return ops.ab_step(x, deis_coef, new_eps, eps_pred)
