"""CPU checks of the oracle's restatement of the L-parametrised and rotated-frame CLD samplers (oracle/cld_ext.py) and of the
product's vectorised host math for them (gddim_amd/cld/sde_lib.py: LSDE, MLCLD)."""
import numpy as np
import pytest

from oracle import cld as ocld
from oracle import cld_ext as ox

N_SCAN = 20000  # RK4 steps of the psi2 table in these tests (the samplers use 100 000; RK4 is converged long before)


@pytest.fixture(scope="module")
def osde():
    return ocld.CLD(is_R_rk=True, R_dt=1e-5, dtype=np.float64)


@pytest.fixture(scope="module")
def oml(osde):
    return ox.MLCLD(osde, N=N_SCAN)


def test_psi2_scan_is_the_frame_change_of_the_matrix_exponential(osde, oml):
    """x = psi1(t) y and y(t) = psi2(t) y(0) describe the flow expm(int_0^t F): psi2(t) = psi1(t)^-1 Psi(0, t)."""
    for t in (1e-3, 0.05, 0.37, 1.0):
        closed = ox.s_inv_psi1(osde, t) @ osde.s_psi(0.0, t)
        assert np.abs(oml.s_psi2_fn(np.float64(t)) - closed).max() < 1e-9
        # psi1 is the exponential of F1 alone
        assert np.abs(ox.s_psi1(osde, t) @ ox.s_inv_psi1(osde, t) - np.eye(2)).max() < 1e-14
    # transition in the rotated frame composed with the frame changes is the plain transition
    s, t = 0.8, 0.3
    assert np.abs(ox.s_psi1(osde, t) @ oml.s_psi(s, t) @ ox.s_inv_psi1(osde, s) - osde.s_psi(s, t)).max() < 1e-8


def test_batched_forms_equal_scalar_forms(osde, oml):
    ts = np.array([1e-3, 0.02, 0.5, 0.97])
    L = ox.LSDE(osde)
    assert np.abs(L.v_eps_integrand(ts) - np.stack([L.s_eps_integrand(t) for t in ts])).max() < 1e-10
    assert np.abs(oml.v_eps_integrand(ts) - np.stack([oml.s_eps_integrand(t) for t in ts])).max() < 1e-10
    assert np.abs(oml.vs_psi(ts, 0.3) - np.stack([oml.s_psi(s, 0.3) for s in ts])).max() < 1e-12


def test_cholesky_parametrisation(osde):
    L = ox.LSDE(osde)
    for t in (1e-3, 0.2, 1.0):
        R = osde.s_R(np.float64(t))
        l = L.s_L(t)
        assert np.abs(l @ l.T - R @ R.T).max() < 1e-14 and l[0, 1] == 0
        E = l.T @ ocld.inv_2x2(R.T)  # epsR2epsL: an orthogonal map (both L and R are square roots of the covariance)
        assert np.abs(E @ E.T - np.eye(2)).max() < 1e-9
        # 1/2 G G L^-T E = 1/2 G G R^-T: the two parametrisations integrate the same score term
        assert np.abs(L.s_eps_integrand(t) @ E - osde.s_eps_integrand(t)).max() < 1e-9 * np.abs(osde.s_eps_integrand(t)).max()


def test_product_host_math_matches_oracle(native_lib, osde):
    from gddim_amd.cld import sde_lib
    sde = sde_lib.CLD(is_R_rk=True, R_dt=1e-5, x64=True)
    rev = ocld.get_rev_ts(osde, 2, 5)
    pl, ol = sde_lib.LSDE(sde), ox.LSDE(osde)
    cl, co = pl.get_deis_coef(2, rev), ol.get_deis_coef(2, rev)
    assert cl.shape == (5, 5, 2, 2) and np.abs(cl - co).max() < 1e-10 * np.abs(co).max()
    for t in rev[:-1]:
        want = ol.s_L(t).T @ ocld.inv_2x2(osde.s_R(np.float64(t)).T)
        assert np.abs(pl.epsR2epsL_matrix(t) - want).max() < 1e-12
    pm, om = sde_lib.MLCLD(sde), ox.MLCLD(osde, N=N_SCAN)
    for t in (1e-3, 0.4, 1.0):
        assert np.abs(pm.s_psi2_fn(t) - om.s_psi2_fn(np.float64(t))).max() < 1e-8
        assert np.abs(pm.y2x_matrix(t) - ox.s_psi1(osde, t)).max() == 0 and np.abs(pm.x2y_matrix(t) - ox.s_inv_psi1(osde, t)).max() == 0
    cm, co = pm.get_deis_coef(1, rev[:4]), om.get_deis_coef(1, rev[:4])
    assert np.abs(cm - co).max() < 1e-7 * np.abs(co).max()
    m = sde.mixed_u_matrix(0.3)
    assert np.abs(m[:, 1] - sde.s_invR(0.3)[:, 1]).max() == 0 and np.abs(m[:, 0]).max() == 0


def _dirac(osde):
    c = 0.3
    z = np.random.default_rng(0).standard_normal((5, 1, 1, 1, 2))

    def mu(t):
        return osde.s_psi(0.0, t) @ np.array([c, 0.0])

    def eps_fn(u, tvec):
        t = float(tvec[0])
        return np.einsum("ij,b...j->b...i", ocld.inv_2x2(osde.s_R(t)), u - mu(t))

    u0 = mu(1.0) + np.einsum("ij,b...j->b...i", osde.s_R(1.0), z)
    want = mu(1e-3) + np.einsum("ij,b...j->b...i", osde.s_R(1e-3), z)
    return eps_fn, u0, want


def test_dirac_data_trajectories(osde, oml):
    """With the exact eps of Dirac data every sampler must land on mu_eps + R_eps z up to its quadrature error; the rotated
    frame additionally carries the reference's own psi1(eps/2)-instead-of-psi1(eps) frame change (an O(eps) rotation)."""
    eps_fn, u0, want = _dirac(osde)
    # The ODE's R(t) is a square root that rotates quickly as t -> 0 while chol(R R^T) does not, so the exact eps_R = z of this
    # problem turns into a rotating eps_L = E(t) z: the L-parametrised extrapolation needs more steps here (it is aimed at
    # networks whose output is smoother in the L frame) but converges to the same point.
    err = []
    for nfe in (25, 100):
        x, v, n = ox.ldeis_sampler(osde, eps_fn, u0, nfe, 2)
        err.append(max(np.abs(x - want[..., 0]).max(), np.abs(v - want[..., 1]).max()))
    assert n == 100 and err[1] < 5e-3 and err[1] < 0.1 * err[0]
    x, v, n = ox.mldeis_sampler(osde, eps_fn, u0, 12, 1, mlsde=oml)
    assert np.abs(x - want[..., 0]).max() < 2e-2 and np.abs(v - want[..., 1]).max() < 2e-2
    # undoing the frame quirk leaves only quadrature error
    u = np.stack([x, v], axis=-1)
    fix = ox.s_psi1(osde, 1e-3) @ ox.s_inv_psi1(osde, 0.5e-3)
    u = np.einsum("ij,b...j->b...i", fix, u)
    assert np.abs(u - want).max() < 6e-3


def test_mixed_score_is_a_reparametrisation(osde):
    """A network trained with mixed_score predicts eps - invR(t) [0, v]; wrapping it restores eps exactly."""
    eps_fn, u0, _ = _dirac(osde)

    def net(u, tvec):
        x0 = u.copy()
        x0[..., 0] = 0.0
        inv_r = ocld.inv_2x2(osde.s_R(np.float64(tvec[0])))
        return eps_fn(u, tvec) - np.einsum("ij,b...j->b...i", inv_r, x0)

    a = ocld.deis_sampler(osde, eps_fn, u0, 8, 2, denoising=True)
    b = ocld.deis_sampler(osde, ox.mixed_eps_fn(osde, net), u0, 8, 2, denoising=True)
    assert np.abs(a[0] - b[0]).max() < 1e-9 and np.abs(a[1] - b[1]).max() < 1e-9


def test_sscs_ou_half_step_is_the_exact_ou_transition(osde):
    """The OU part of the splitting (drift F u, diffusion G in the time convention t' = 1 - t) has the closed-form transition the
    sampler hard-codes: mean expm(-F b) and the covariance solving the Lyapunov equation; checked against scipy."""
    import scipy.linalg
    s_t, s_next = 0.3, 0.35
    u = np.random.default_rng(0).standard_normal((3, 1, 2))
    zero = np.zeros_like(u)
    mean, cov = ox.sscs_ou(osde, u, s_t, s_next, zero)
    b = -(osde._beta_int(1 - s_next) - osde._beta_int(1 - s_t))  # integral of beta over the half step
    # sampling-direction OU: du = beta [[0, -m_inv], [1, -Gamma m_inv]] u dt + sqrt(2 Gamma beta) dW_v  (CLD paper eq. 2, reversed x<->drift sign)
    A = np.array([[0.0, -float(osde.m_inv)], [1.0, -float(osde.Gamma * osde.m_inv)]])
    want_mean = np.einsum("ij,b...j->b...i", scipy.linalg.expm(A * b), u)
    assert np.abs(mean - want_mean).max() < 1e-12
    # covariance: int_0^b expm(A s) diag(0, 2 Gamma) expm(A s)^T ds by adaptive quadrature
    import scipy.integrate
    Q = np.array([[0.0, 0.0], [0.0, 2 * float(osde.Gamma)]])
    want_cov, _ = scipy.integrate.quad_vec(lambda s: scipy.linalg.expm(A * s) @ Q @ scipy.linalg.expm(A * s).T, 0.0, b, epsabs=1e-13)
    assert np.abs(cov - want_cov).max() < 1e-10
    assert np.all(np.linalg.eigvalsh(cov) > 0)


def test_product_sscs_and_em_maps_match_oracle(native_lib, osde):
    from gddim_amd.cld import sampling, sde_lib
    sde = sde_lib.CLD(is_R_rk=True, R_dt=1e-5, x64=True)
    M, cov = sampling.sscs_ou(sde, 0.3, 0.35)
    _, ocov = ox.sscs_ou(osde, np.zeros((1, 1, 2)), 0.3, 0.35, np.zeros((1, 1, 2)))
    assert np.abs(cov - ocov).max() < 1e-15
    S = sampling.noise_factor(cov)
    assert np.abs(S @ S.T - cov).max() < 1e-15 and S[0, 1] == 0
    bad = np.array([[1.0, 1.0], [1.0, 1.0 - 1e-17]])  # singular up to round-off: falls back to the symmetric root
    Sb = sampling.noise_factor(bad)
    assert np.abs(Sb @ Sb.T - bad).max() < 1e-12


# ---- the lambda-family reverse SDE (sdeis) --------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def olam(osde):
    # dt = 1e-5 like the reference (~20 s of python RK4): hat_F is stiff near t = 0 (Sigma^-1 blows up) and the scan's grid
    # linspace(0, 1 + dt, int(1/dt) + 1, endpoint=False) has spacing 1.00001e-5 for steps of 1e-5 -- both are part of its results
    return ox.LambdaSDE(osde, 1.0, use_order0=True, dt=1e-5)


def test_lambda_sde_reverse_transition_keeps_the_marginal(osde, olam):
    """For Dirac data at 0 the reverse SDE is linear, du = hat_F u dt + lambda G dw, and must carry N(0, Sigma_s) into N(0, Sigma_t):
    hat_psi Sigma_s hat_psi^T + P = Sigma_t with dP/dtau = hat_F P + P hat_F^T - lambda^2 G G^T.  That pins hat_psi.  The reference's
    cond_rev_cov integrates hat_F P + P hat_F instead (no transpose, contrary to its own comment): its result is not symmetric,
    which this test records; the restatement follows the code."""
    s, t = 0.5, 0.4
    hp = olam.s_hat_psi(s, t)

    def with_transpose(n_step=1000):
        dt, val, tt = (t - s) / n_step, np.zeros((2, 2)), s
        fn = lambda _x, _t: olam.s_hat_F(_t) @ _x + _x @ olam.s_hat_F(_t).T - osde.s_G(_t) @ osde.s_G(_t).T
        for _ in range(n_step):
            val = ox.runge_kutta(val, tt, dt, fn)
            tt += dt
        return val

    assert np.abs(hp @ olam.s_cov(s) @ hp.T + with_transpose() - olam.s_cov(t)).max() < 1e-5
    as_coded = olam.cond_rev_cov(s, t, n_step=1000)
    assert abs(as_coded[0, 1] - as_coded[1, 0]) > 0.05  # the quirk
    # lambda = 0: no noise at all, and hat_psi is the probability-flow transition of the same linear problem
    l0 = ox.LambdaSDE(osde, 0.0, dt=1e-3)
    assert np.abs(l0.cond_rev_cov(s, t, n_step=50)).max() == 0


def test_product_lambda_sde_matches_oracle(native_lib, osde, olam):
    from gddim_amd.cld import sampling, sde_lib
    sde = sde_lib.CLD(is_R_rk=True, R_dt=1e-5, x64=True)
    pl = sde_lib.LambdaSDE(sde, 1.0, use_order0=True)
    for s, t in ((1.0, 0.7), (0.3, 0.1), (0.01, 0.001)):
        assert np.abs(pl.s_hat_psi(s, t) - olam.s_hat_psi(s, t)).max() < 1e-6 * max(1.0, np.abs(olam.s_hat_psi(s, t)).max())
        assert np.abs(pl.s_hat_F(t) - olam.s_hat_F(t)).max() < 1e-9 * np.abs(olam.s_hat_F(t)).max()
        a, b = pl.cond_rev_cov(s, t, n_step=400), olam.cond_rev_cov(s, t, n_step=400)
        assert np.abs(a - b).max() < 1e-9 * max(1.0, np.abs(b).max())
    rev = ocld.get_rev_ts(osde, 2, 3)
    pc, oc = pl.get_poly_eps_coef(1, rev), olam.get_poly_eps_coef(1, rev)
    assert pc.shape == (3, 3, 2, 2) and np.abs(pc - oc).max() < 1e-5 * np.abs(oc).max()
    u0 = pl.update_coef(rev[0], rev[1])
    want_eps = (olam.s_hat_psi(rev[0], rev[1]) - osde.s_psi(rev[0], rev[1])) @ osde.s_R(np.float64(rev[0]))
    assert np.abs(u0[0] - osde.s_psi(rev[0], rev[1])).max() < 1e-12 and np.abs(u0[1] - want_eps).max() < 1e-6
    # table layouts: order 0 with use_order0 -> [Psi, eps, 0, cov]; order q otherwise -> [Psi, q + 2 eps slots, cov]
    assert pl.get_deis_coef(0, rev).shape == (3, 4, 2, 2) and np.abs(pl.get_deis_coef(0, rev)[:, 2]).max() == 0
    assert pl.get_deis_coef(1, rev).shape == (3, 5, 2, 2)
    f = sampling.svd_noise_factor(np.array([[2.0, 0.3], [0.3, 1.0]]))
    assert np.abs(f @ f.T - np.array([[2.0, 0.3], [0.3, 1.0]])).max() < 1e-12  # a root when the matrix is a covariance
    assert np.abs(f - ox.svd_factor(np.array([[2.0, 0.3], [0.3, 1.0]]))).max() == 0


def test_hybdeis_grid_and_table(native_lib, osde):
    """`hybdeis` (cld_jax/sampling.py:255-269): the grid is int(num_step * 0.3) uniform knots from T towards 0.3 T, then the
    power-law grid -- which restarts at T (kept literally).  The product's table on that non-monotone grid equals the oracle's."""
    from gddim_amd.cld import sampling, sde_lib
    rev = ox.hyd_rev_ts(osde, 11, 0.3, 0.3, 2.0, denoising=True)  # num_step 10: 3 noise knots + 8 image knots
    assert rev.shape == (11,) and np.allclose(rev[:3], [1.0, 1.0 - 0.7 / 3, 1.0 - 1.4 / 3]) and rev[3] == 1.0 and abs(rev[-1] - 1e-3) < 1e-15
    assert np.all(np.diff(rev[3:]) < 0) and rev[2] < rev[3]  # the restart
    sde = sde_lib.CLD(is_R_rk=True, R_dt=1e-5, x64=True)
    seen = {}
    orig = sampling._impl_deis_sampler
    try:
        sampling._impl_deis_sampler = lambda sde_, model, shape, nfe, inv, order, rev_ts, denoising, is_p: seen.update(rev=np.asarray(rev_ts), order=order)
        sampling.get_hyd_deis_sampler(sde, None, (8, 8, 3), 11, None, 2, noise_nfe_ratio=0.3, img_t_ratio=0.3, ts_order=2.0, denoising=True)
    finally:
        sampling._impl_deis_sampler = orig
    assert np.abs(seen["rev"] - rev).max() < 1e-15 and seen["order"] == 2
    got, want = sde.get_deis_coef(1, rev[:6]), osde.get_deis_coef(1, rev[:6])
    assert np.abs(got - want).max() < 1e-9 * np.abs(want).max()


def test_naive_coef_is_the_euler_step(native_lib, osde):
    """prepare_naive_coef (cld_jax/sde_lib.py:276-287, `order0` with is_em=True): product == oracle, and both are the first-order
    expansion of the exponential-integrator coefficients (gap O(dt^2))."""
    from gddim_amd.cld import sde_lib
    sde = sde_lib.CLD(is_R_rk=True, R_dt=1e-5, x64=True)
    rev = ocld.get_rev_ts(osde, 2, 200)
    pm, pe = sde.prepare_naive_coef(rev)
    om, oe = osde.prepare_naive_coef(rev)
    assert pm.shape == (200, 2, 2) and np.abs(pm - om).max() < 1e-12 and np.abs(pe - oe).max() < 1e-9 * np.abs(oe).max()
    dt = rev[1] - rev[0]
    assert np.abs(pm[0] - (np.eye(2) + np.array([[0.0, 16.0], [-4.0, -16.0]]) * dt)).max() < 1e-14
    em, ee = sde.prepare_order0_coef(rev)
    i = 150  # a short step
    h = abs(rev[i + 1] - rev[i])
    assert np.abs(pm[i] - em[i]).max() < 200 * h * h and np.abs(pe[i] - ee[i]).max() < 0.1 * np.abs(ee[i]).max()


def test_variants_run_on_the_float32_base(native_lib):
    """With config.model.x64=False the base CLD's members are float32 (the reference's arithmetic); the variant wrappers read R(t) from it
    and build their own tables in float64.  They must construct and stay within the float32-vs-float64 spread of the base tables."""
    from gddim_amd.cld import sampling, sde_lib
    s32, s64 = sde_lib.CLD(is_R_rk=True, R_dt=1e-5), sde_lib.CLD(is_R_rk=True, R_dt=1e-5, x64=True)
    assert s32.s_R(0.3).dtype == np.float32 and s64.s_R(0.3).dtype == np.float64
    rev32, rev64 = sampling.get_rev_ts(s32, 2, 5), sampling.get_rev_ts(s64, 2, 5)
    assert rev32.dtype == np.float32 and np.abs(rev32 - rev64).max() < 1e-6

    def close(a, b, rel):
        return np.isfinite(a).all() and np.abs(a - b).max() <= rel * max(1.0, np.abs(b).max())

    for make, order, rel in ((sde_lib.LSDE, 1, 1e-3), (sde_lib.MLCLD, 1, 1e-3), (lambda s: sde_lib.LambdaSDE(s, 1.0, True), 0, 1e-3),
                             (lambda s: sde_lib.LambdaSDE(s, 1.0, True), 1, 1e-3)):
        a, b = make(s32).get_deis_coef(order, rev32), make(s64).get_deis_coef(order, rev64)
        assert a.shape == b.shape and close(np.asarray(a, dtype=np.float64), b, rel), (order, np.abs(a - b).max())
    c32, n32 = sampling._em_coef(s32, rev32, 1.0)
    c64, n64 = sampling._em_coef(s64, rev64, 1.0)
    assert close(c32, c64, 1e-3) and close(n32, n64, 1e-3)
    assert close(sampling.sscs_ou(s32, 0.3, 0.35)[1], sampling.sscs_ou(s64, 0.3, 0.35)[1], 1e-12)  # formed in float64 in both modes
