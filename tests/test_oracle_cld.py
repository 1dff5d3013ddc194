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
    # the reference computes this table in float32 (x64=False): 5.7e-5 absolute on entries up to 16.6 at NFE 10
    assert np.abs(c32 - c64).max() < 1e-4 * np.abs(c64).max()
    # ... and the order in which float32 accumulates the 10 000 terms of each entry (XLA's, not in the source) is worth half of that
    c32s = ocld.CLD(is_R_rk=True, R_dt=1e-6, dtype=np.float32, sum_order="sequential").get_deis_coef(2, ocld.get_rev_ts(o32, 2, 9))
    d_order = np.abs(c32s.astype(np.float64) - c32).max()
    print("NFE 10 table: float32 vs float64 %.2e, float32 pairwise vs sequential sum %.2e (scale %.1f)"
          % (np.abs(c32 - c64).max(), d_order, np.abs(c64).max()))
    assert 1e-6 < d_order < 1e-4


# |x_hip - x_float64 oracle| measured on the MI355X at full size with the SAME tables on both sides (near-denoiser construction of
# tests/test_gpu_parity_abs.py; profiles/r03_parity_abs_tests.log, profiles/r04_parity_abs_tests.log): the kernels' share of the budget
HIP_ERR = {10: 6.8e-5, 20: 3.7e-5, 50: 1.4e-5, 100: 9.4e-6}
TABLE_SPREAD = {}


@pytest.mark.parametrize("nfe", [10, 20, 50, 100])
def test_parity_budget_float32_tables_end_to_end(osde, nfe):
    """The reference runs x64=False: float32 R scan, float32 10 000-term Riemann sums, float32 time grid (cld_jax/sde_lib.py:93-118,
    308-319, deis.py:19-59, sampling.py:241-249).  How far does that arithmetic -- in either accumulation order XLA may use -- move
    the SAMPLE away from exact (float64) tables?  Same model, same noise, float64 loop, near-denoiser (bounded trajectory): only the
    tables differ.  This is the largest known deviation between any two faithful executions of the reference's source, larger than
    the HIP kernels' own error; the north-star's 1e-3 per pixel has to hold for the sum of the two."""
    import copy
    import torch
    from conftest import tiny_config
    from oracle import ncsnpp as onet
    cfg = tiny_config("cld")
    params = onet.random_params(cfg, 6, seed=3)
    params = copy.deepcopy(params)
    params["Conv_1"]["kernel"] = (params["Conv_1"]["kernel"] * 0.3).astype(np.float32)
    params["Conv_1"]["bias"] = (params["Conv_1"]["bias"] * 0.3).astype(np.float32)
    net64 = onet.cld_eps_fn(params, cfg, dtype=torch.float64)
    s_data = 0.5

    def x_of(t):  # exact eps map of Gaussian data N(0, s^2 I): the "checkpoint" is the same function for every table variant
        R, P = osde.s_R(np.float64(t)), osde.s_psi(0.0, np.float64(t))
        return R.T @ np.linalg.inv(P @ np.diag([s_data ** 2, 0.0]) @ P.T + R @ R.T)

    def eps(u, t):
        return net64(u, t) + ocld.bmm(np.stack([x_of(tb) for tb in t]), u)

    def lin(u, t):
        return ocld.bmm(np.stack([x_of(tb) for tb in t]), u)

    rng = np.random.default_rng(7)
    shape = (2, cfg.data.image_size, cfg.data.image_size, 3)
    u = np.stack([rng.standard_normal(shape), rng.standard_normal(shape) / 2], axis=-1)

    def run(sde, fn=eps):
        return ocld.deis_sampler(sde, fn, u, nfe, 2, ts_order=2, denoising=True, inverse_scaler=ocld.inverse_scaler_centered)[0]

    x64 = run(osde)
    moved = np.abs(x64 - run(osde, lin)).max()
    spread = {}
    for order in ("pairwise", "sequential"):
        o32 = ocld.CLD(is_R_rk=True, R_dt=1e-6, dtype=np.float32, sum_order=order)
        spread[order] = float(np.abs(run(o32) - x64).max())
    TABLE_SPREAD[nfe] = spread
    print("NFE %3d: float32-table sample vs float64-table sample: max|dx| %.2e (pairwise sum) %.2e (sequential sum); HIP kernels %.1e; "
          "x in [%.2f, %.2f], network moved x by %.2f" % (nfe, spread["pairwise"], spread["sequential"], HIP_ERR[nfe], x64.min(), x64.max(), moved))
    assert moved > 0.02 and np.abs(x64).max() < 50.0  # the network matters and the trajectory is bounded
    assert max(spread.values()) > 1e-6                # the tables' arithmetic is visible in the sample
    assert max(spread.values()) + HIP_ERR[nfe] <= 1e-3, "float32-table spread + HIP error exceeds the north-star's 1e-3 per pixel"


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
    """gddim_amd.cld.sde_lib / deis (vectorised, R table from the C ABI) in float64 mode (config.model.x64=True) vs the literal oracle."""
    from gddim_amd.cld import sampling, sde_lib
    sde = sde_lib.CLD(is_R_rk=True, R_dt=1e-6, x64=True)
    for t in (1e-3, 0.1, 0.5, 1.0):
        assert np.abs(sde.s_R(t) - osde.s_R(t)).max() < 1e-12
        assert np.abs(sde.s_F(t) - osde.s_F(t)).max() == 0 and np.abs(sde.s_G(t) - osde.s_G(t)).max() < 1e-15
    for nfe, order in ((10, 2), (20, 1), (7, 3)):
        rev = sampling.get_rev_ts(sde, 2, nfe - 1)
        assert rev.dtype == np.float64 and np.abs(rev - ocld.get_rev_ts(osde, 2, nfe - 1)).max() < 1e-15
        assert np.abs(sde.get_deis_coef(order, rev) - osde.get_deis_coef(order, rev)).max() < 1e-11
    m, e = sde.prepare_order0_coef(rev)
    mo, eo = osde.prepare_order0_coef(rev)
    assert np.abs(m - mo).max() < 1e-13 and np.abs(e - eo).max() < 1e-11
    e_sde = sde_lib.CLD(is_R_rk=False, R_dt=1e-5, x64=True)
    e_o = ocld.CLD(is_R_rk=False, R_dt=1e-5, dtype=np.float64)
    assert np.abs(e_sde.s_R(0.3) - e_o.s_R(0.3)).max() < 1e-12


def test_product_honours_x64_false_in_the_references_own_arithmetic(native_lib, osde):
    """config.model.x64=False is the reference's default and means float32 tables (cld_jax/sde_lib.py:112,314 assert it).  The product's
    float32 mode -- native float32 RK4 scan, float32 knots / interpolation / Psi / integrand / time grids / Riemann sums -- against the
    oracle's float32 restatement: the 10^6-step scan and everything elementwise bit for bit, the sums to float32 round-off."""
    from gddim_amd.cld import sampling, sde_lib
    from gddim_amd.configs import cld_accr_dcifar10
    cfg = cld_accr_dcifar10.get_config()
    assert cfg.model.x64 is False
    sde = sde_lib.from_config(cfg)
    assert sde.x64 is False and sde._fp.dtype == np.float32 and sde._xp.dtype == np.float32
    o32 = ocld.CLD(is_R_rk=cfg.model.is_R_rk, R_dt=cfg.model.R_dt, dtype=np.float32)
    idx = ocld.jnp_linspace_int(0, o32.R_table.shape[0] - 1, 100_000, dtype=np.float32)
    assert idx[9999] == 99991  # float32 picks a different knot row than exact arithmetic would (99 990)
    # the product pins its last knot to the table's last row; the literal formula lands there by itself for both scan grids
    # of the shipped configs (R_dt = 1e-6: 1 000 001 rows, R_dt = 1e-5: 100 000 rows) in both dtypes, so the pin is a no-op where tested
    for rows in (1_000_001, 100_000):
        for ft in (np.float32, np.float64):
            lit = ocld.jnp_linspace_int(0, rows - 1, 100_000, dtype=ft)
            assert lit[-1] == rows - 1 and np.array_equal(sde_lib.linspace_int(0, rows - 1, 100_000, ft=ft), lit)
    assert sde_lib.linspace_int(0, 7, 4, ft=np.float32)[-1] == 7
    assert np.array_equal(sde._fp, o32.R_table[idx]) and np.array_equal(sde._xp, o32.R_table_ts[idx])  # the scan, bit for bit
    for t in (1e-3, 0.0123, 0.1, 0.5, 0.99999, 1.0):
        assert np.array_equal(sde.s_R(t), o32.s_R(np.float32(t)))
        assert np.array_equal(sde.s_F(t), o32.s_F(t)) and np.array_equal(sde.s_G(t), o32.s_G(t))
        assert np.array_equal(sde.s_eps_integrand(t), o32.s_eps_integrand(t))
        assert np.array_equal(sde.s_psi(t, 0.5 * t), o32.s_psi(t, 0.5 * t))
    for nfe, order in ((10, 2), (20, 2), (50, 2), (20, 1), (7, 3)):
        rev = sampling.get_rev_ts(sde, 2, nfe - 1)
        assert rev.dtype == np.float32 and np.array_equal(rev, ocld.get_rev_ts(o32, 2, nfe - 1))
        got, want = sde.get_deis_coef(order, rev), o32.get_deis_coef(order, rev)
        assert got.dtype == np.float32
        assert np.array_equal(got[:, 0], want[:, 0])  # Psi_i: elementwise
        # the sums: same terms, same pairwise order up to the grouping of the Lagrange factors -- a few ulp of the entry's scale
        assert np.abs(got.astype(np.float64) - want).max() <= 4 * np.finfo(np.float32).eps * np.abs(want).max(), (nfe, order)
        # and the float32 table sits where float32 arithmetic puts it, not where float64 does
        assert np.abs(got - osde.get_deis_coef(order, ocld.get_rev_ts(osde, 2, nfe - 1))).max() > 1e-6
    m, e = sde.prepare_order0_coef(rev)
    mo, eo = o32.prepare_order0_coef(rev)
    assert np.array_equal(m, mo) and np.abs(e - eo).max() <= 4 * np.finfo(np.float32).eps * np.abs(eo).max()
    m, e = sde.prepare_naive_coef(rev)
    mo, eo = o32.prepare_naive_coef(rev)
    assert np.array_equal(m, mo) and np.array_equal(e, eo)
    # float64 stays available, and the two modes do not share a cache entry
    s64 = sde_lib.CLD(is_R_rk=cfg.model.is_R_rk, R_dt=cfg.model.R_dt, x64=True)
    assert s64._fp.dtype == np.float64 and np.abs(s64.s_R(0.5) - osde.s_R(0.5)).max() < 1e-12
