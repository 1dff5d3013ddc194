"""Known-answer tests that pin the CPU oracle of the CLD sampler math (the reference ships no golden vectors and
cannot run here -- SURVEY.md section 8c), plus the product's host-side coefficient code against the oracle."""
import numpy as np
import pytest
import scipy.integrate
import scipy.linalg

from oracle import cld as ocld


@pytest.fixture(scope="module")
def osde():
    return ocld.CLD(is_R_rk=True, R_dt=1e-6, dtype=np.float64)


def test_psi_is_the_matrix_exponential(osde):
    F = osde.s_F(0.0)
    for s, t in [(0.0, 0.3), (1.0, 0.796), (0.5, 0.1), (0.02, 0.001)]:
        want = scipy.linalg.expm(F * (t - s))  # beta constant: int_s^t F = F (t - s)
        assert np.abs(osde.s_psi(s, t) - want).max() < 1e-12 * max(1.0, np.abs(want).max())


def test_R_is_a_square_root_of_the_lyapunov_covariance(osde):
    F, G = osde.s_F(0.0), osde.s_G(0.0)
    S0 = osde.R_0 @ osde.R_0.T

    def rhs(t, s):
        S = s.reshape(2, 2)
        return (F @ S + S @ F.T + G @ G.T).ravel()

    sol = scipy.integrate.solve_ivp(rhs, (0, 1.0), S0.ravel(), rtol=1e-11, atol=1e-13, dense_output=True)
    for t in (1e-3, 0.1, 0.5, 1.0):
        R = osde.s_R(t)
        assert np.abs(R @ R.T - sol.sol(t).reshape(2, 2)).max() < 2e-6
    # SURVEY.md B.4 scratch anchors
    assert np.allclose(osde.s_R(1.0), [[0.9966196, -0.08205698], [0.04103561, 0.4983101]], atol=1e-6)
    assert np.allclose(osde.s_R(1e-3), [[8.4875e-4, 1.86379e-3], [-2.673631e-2, 1.2978354e-1]], atol=1e-7)


def test_float32_scan_tracks_float64(osde):
    o32 = ocld.CLD(is_R_rk=True, R_dt=1e-6, dtype=np.float32)
    for t in (1e-3, 0.1, 0.5, 1.0):
        assert np.abs(o32.s_R(t).astype(np.float64) - osde.s_R(t)).max() < 2e-5
    rev = ocld.get_rev_ts(osde, 2, 9)
    c64 = osde.get_deis_coef(2, rev)
    c32 = o32.get_deis_coef(2, ocld.get_rev_ts(o32, 2, 9))
    # the reference computes this table in float32: the float64 table used by the product is within 1e-4 of it
    assert np.abs(c32 - c64).max() < 1e-4 * np.abs(c64).max()


def test_euler_scan_quirk_grid(osde):
    """R_dt=1e-5 (default config): int(1/1e-5) == 99999, so the grid has 100000 points (sde_lib.py:114)."""
    o = ocld.CLD(is_R_rk=False, R_dt=1e-5, dtype=np.float64)
    assert o.R_table.shape[0] == 100000
    # first-order scheme on the stiff early dynamics (R_0 ~ 1e-3) with the grid quirk vs RK4 at 1e-6: ~6e-3 apart
    assert np.abs(o.s_R(0.5) - osde.s_R(0.5)).max() < 2e-2


def test_rev_ts_and_deis_table_shape(osde):
    rev = ocld.get_rev_ts(osde, 2, 9)
    assert np.allclose(rev, [1, .796382, .615919, .458610, .324456, .213456, .125610, .060919, .019382, .001], atol=1e-6)
    c = osde.get_deis_coef(2, rev)
    assert c.shape == (9, 5, 2, 2)
    assert np.all(c[:, 4] == 0)            # highest_order = order + 1 pads one all-zero slot
    assert np.all(c[0, 2:] == 0) and np.all(c[1, 3:] == 0)  # warm start: row i < q has order i
    assert np.allclose(c[0, 0], [[-3.206645, -16.610243], [4.152561, 13.403598]], atol=1e-5)


def test_deis_coefficients_match_adaptive_quadrature_and_row_sum(osde):
    rev = ocld.get_rev_ts(osde, 2, 9)
    c = osde.get_deis_coef(2, rev)
    c0 = osde.get_deis_coef(0, rev)
    # Lagrange basis sums to one => sum over slots == the order-0 coefficient (same left-Riemann rule)
    assert np.abs(c[:, 1:].sum(axis=1) - c0[:, 1]).max() < 1e-10

    def integrand(tau, i, j, nodes):
        w = 1.0
        for k in range(len(nodes)):
            if k != j:
                w *= (tau - nodes[k]) / (nodes[j] - nodes[k])
        return (osde.s_psi(tau, rev[i + 1]) @ osde.s_eps_integrand(tau)) * w

    for i in (2, 8):
        nodes = rev[i - 2:i + 1]
        for slot in range(3):
            exact, _ = scipy.integrate.quad_vec(lambda tau: integrand(tau, i, 2 - slot, nodes), rev[i], rev[i + 1], epsrel=1e-8)
            # the reference's rule is first order: relative gap ~ 1/num_item -- must NOT be "fixed"
            gap = np.abs(c[i, 1 + slot] - exact).max() / np.abs(exact).max()
            assert 1e-7 < gap < 2e-3, (i, slot, gap)


def test_dirac_data_trajectory(osde):
    """Data x0 == c: eps*(u,t) = R_t^-1 (u - Psi(0,t)[c,0]^T) is exact and u_t = mu_t + R_t z; the oracle's deis
    loop (order 2, NFE 20) must land within the quadrature error of the closed form, and the order-0 loop too."""
    c = 0.3
    z = np.random.default_rng(0).standard_normal((5, 1, 1, 1, 2))

    def mu(t):
        return osde.s_psi(0.0, t) @ np.array([c, 0.0])

    def eps_fn(u, tvec):
        t = float(tvec[0])
        return np.einsum("ij,b...j->b...i", ocld.inv_2x2(osde.s_R(t)), u - mu(t))

    u0 = mu(1.0) + np.einsum("ij,b...j->b...i", osde.s_R(1.0), z)
    for order, tol in ((2, 5e-3), (1, 5e-3), (0, 5e-3)):
        x, v, n = ocld.deis_sampler(osde, eps_fn, u0, 20, order, ts_order=2, denoising=False)
        want = mu(1e-3) + np.einsum("ij,b...j->b...i", osde.s_R(1e-3), z)
        assert np.abs(x - want[..., 0]).max() < tol and np.abs(v - want[..., 1]).max() < tol, order
    x0, v0, _ = ocld.order0_sampler(osde, eps_fn, u0, 20, denoising=False)
    assert np.abs(x0 - want[..., 0]).max() < 5e-3


def test_multistep_ab_step_semantics():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 3, 2))
    new = rng.standard_normal((2, 3, 2))
    hist = rng.standard_normal((2, 2, 3, 2))
    coef = rng.standard_normal((4, 2, 2))
    out, h2 = ocld.multistep_ab_step(x, coef, new, hist)
    want = x @ coef[0].T + new @ coef[1].T + hist[0] @ coef[2].T + hist[1] @ coef[3].T
    assert np.allclose(out, want)
    assert np.array_equal(h2[0], new) and np.array_equal(h2[1], hist[0])


def test_product_host_math_matches_oracle(native_lib, osde):
    """gddim_amd.cld.sde_lib / deis (vectorised float64, R table from the C ABI) vs the literal oracle."""
    from gddim_amd.cld import sampling, sde_lib
    sde = sde_lib.CLD(is_R_rk=True, R_dt=1e-6)
    for t in (1e-3, 0.1, 0.5, 1.0):
        assert np.abs(sde.s_R(t) - osde.s_R(t)).max() < 1e-12
        assert np.abs(sde.s_F(t) - osde.s_F(t)).max() == 0 and np.abs(sde.s_G(t) - osde.s_G(t)).max() < 1e-15
    for nfe, order in ((10, 2), (20, 1), (7, 3)):
        rev = sampling.get_rev_ts(sde, 2, nfe - 1)
        assert np.abs(rev - ocld.get_rev_ts(osde, 2, nfe - 1)).max() < 1e-15
        assert np.abs(sde.get_deis_coef(order, rev) - osde.get_deis_coef(order, rev)).max() < 1e-11
    m, e = sde.prepare_order0_coef(rev)
    mo, eo = osde.prepare_order0_coef(rev)
    assert np.abs(m - mo).max() < 1e-13 and np.abs(e - eo).max() < 1e-11
    e_sde = sde_lib.CLD(is_R_rk=False, R_dt=1e-5)
    e_o = ocld.CLD(is_R_rk=False, R_dt=1e-5, dtype=np.float64)
    assert np.abs(e_sde.s_R(0.3) - e_o.s_R(0.3)).max() < 1e-12
