"""DEIS Adams-Bashforth coefficients (host) and the fused AB step (HIP).

Replaces cld_jax/deis.py:19-95 (coefficient generator) and :141-151 (multistep_ab_step).

The coefficient of step i (order q, nodes t_{i-q..i}) and slot s is the reference's own quadrature,
    C_{i,s} = sum_{n<10000} Psi(tau_n, t_{i+1}) . 1/2 G G R(tau_n)^-T . l_{q-s}(tau_n) . dtau,
a LEFT Riemann sum on linspace(t_i, t_{i+1}, 10000, endpoint=False) -- kept on purpose: its O(1/n) error is
part of the reference's results.  Rows i < q use order i (warm start), and the table is
zero-padded to highest_order+1 slots.  Unlike the reference, the integrand Psi . g(tau) is evaluated once per
step and shared by the q+1 Lagrange weights (the same numbers: the reference recomputes them per slot).

Arithmetic: the dtype of the SDE object (`sde._ft`: float32 for config.model.x64=False, the reference's default; float64
otherwise; the wrapper SDEs of the sampler variants have none and run in float64).  Every elementwise operation is in the
order of the reference's expression; the 10 000-term sum is numpy's pairwise sum along a contiguous axis, where the reference
has whatever order XLA gives `jnp.sum` (a tree on the GPU backend) -- see DESIGN.md section 2.
"""
import numpy as np

from .. import ops

NUM_ITEM = 10000


def _linspace(start, stop, num, ft):
    # jnp.linspace(start, stop, num, endpoint=False) of jax 0.2.8 in the computation dtype (see sde_lib.linspace)
    step = np.arange(num, dtype=ft) / ft(num)
    return ft(start) * (ft(1) - step) + ft(stop) * step


def _lagrange_weights(t_inter, nodes):
    """(len(nodes), n): l_j(tau) for every node j, as single_poly_coef forms it (deis.py:29-36): prod(t - t_k) / prod(t_j - t_k)
    with the k = j factors set to one."""
    q1 = len(nodes)
    ft = t_inter.dtype.type
    w = np.empty((q1, t_inter.shape[0]), dtype=t_inter.dtype)
    for j in range(q1):
        num = t_inter[:, None] - nodes[None, :]
        den = nodes[j] - nodes
        num[:, j] = ft(1)
        den[j] = ft(1)
        w[j] = np.prod(num, axis=1) / np.prod(den)
    return w


def _sum0(a):
    """Sum over the leading axis, accumulated along a contiguous axis (numpy's pairwise sum) in the array's dtype."""
    return np.sum(np.ascontiguousarray(np.moveaxis(a, 0, -1)), axis=-1)


def get_ab_eps_coef(sde, highest_order, timesteps, order, num_item=NUM_ITEM):
    """(N, highest_order+1, 2, 2); slot 0 multiplies the newest epsilon."""
    ft = np.dtype(getattr(sde, '_ft', np.float64)).type
    ts = np.asarray(timesteps).astype(ft)
    N = len(ts) - 1
    out = np.zeros((N, highest_order + 1, 2, 2), dtype=ft)
    for i in range(N):
        q = min(i, order)  # warm start: the recursion of deis.py:71-95 gives row i order min(i, order)
        t0, t1 = ts[i], ts[i + 1]
        t_inter = _linspace(t0, t1, num_item, ft)
        dt = (t1 - t0) / ft(num_item)
        base = np.einsum("nij,njk->nik", sde.vs_psi(t_inter, t1), sde.v_eps_integrand(t_inter)).astype(ft)  # 2-term dots
        w = _lagrange_weights(t_inter, ts[i - q:i + 1])  # node index j <-> t_{i-q+j}
        for s in range(q + 1):
            out[i, s] = _sum0(base * w[q - s][:, None, None]) * dt
    return out


def get_order0_eps_coef(sde, timesteps, num_item=1000):
    """(N, 2, 2): s_eps_fn of prepare_order0_coef (cld_jax/sde_lib.py:290-296) -- einsum(psi, integrand * dt).sum(axis=0)."""
    ft = np.dtype(getattr(sde, '_ft', np.float64)).type
    ts = np.asarray(timesteps).astype(ft)
    out = np.empty((len(ts) - 1, 2, 2), dtype=ft)
    for i in range(len(ts) - 1):
        t0, t1 = ts[i], ts[i + 1]
        dt = (t1 - t0) / ft(num_item)
        t_inter = _linspace(t0, t1, num_item, ft)
        out[i] = _sum0(np.einsum("nij,njk->nik", sde.vs_psi(t_inter, t1), sde.v_eps_integrand(t_inter) * dt).astype(ft))
    return out


def multistep_ab_step(x, deis_coef, new_eps, eps_pred):
    """x' = Psi x + sum_o C_o full_eps[o], full_eps = [new_eps, *eps_pred]; returns (x', full_eps[:-1])."""
    return ops.ab_step(x, deis_coef, new_eps, eps_pred)
