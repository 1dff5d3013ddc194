"""Known-answer tests for the blur-diffusion oracle and the product's host-side blur schedule."""
import numpy as np
import scipy.fft

from oracle import blur as oblur


def test_dct_construction_is_the_orthonormal_dct2():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 32, 32, 3))
    y = oblur.batch_img_dct(x)
    assert np.abs(y - scipy.fft.dctn(x, type=2, norm="ortho", axes=(1, 2))).max() < 1e-13
    assert np.abs(oblur.batch_img_idct(y) - x).max() < 1e-13
    assert abs(np.sum(x * x) - np.sum(y * y)) < 1e-9 * np.sum(x * x)  # orthonormal: Parseval


def test_schedule_anchors():
    sde = oblur.SDE()
    assert abs(sde.sampling_T - 0.9959798) < 1e-6          # rho2t(80), SURVEY.md App. C
    a = sde.t2alpha_fn(sde.sampling_T)
    assert abs(np.sqrt(1 - a) / np.sqrt(a) - 80.0) < 0.01   # sigma / sqrt(alpha) ~ 80
    D = sde.get_frequency_scaling(np.array([0.5]))
    assert D.shape == (1, 32, 32, 1) and abs(D[0, 0, 0, 0] - 1.0) < 1e-12 and D.min() >= 1e-3
    rev = oblur.get_rev_ts(sde, 2, 50)
    assert rev.shape == (51,) and abs(rev[-1] - 1e-5) < 1e-15 and np.all(np.diff(rev) < 0)


def test_order0_update_is_exact_for_dirac_data():
    sde = oblur.SDE()
    rng = np.random.default_rng(1)
    y0, z = rng.standard_normal((2, 32, 32, 3)), rng.standard_normal((2, 32, 32, 3))
    rev = oblur.get_rev_ts(sde, 2, 5)
    start = sde.y_mean_coef(rev[:1]) * y0 + sde.y_std_coef(rev[:1])[:, None, None, None] * z

    def x_eps_fn(x, t):  # network stand-in whose DCT is exactly z
        return oblur.batch_img_idct(z)

    x, n = oblur.order0_sampler(sde, x_eps_fn, start, 5, ts_order=2)
    want = oblur.batch_img_idct(sde.y_mean_coef(rev[-1:]) * y0 + sde.y_std_coef(rev[-1:])[:, None, None, None] * z)
    assert n == 5 and np.abs(x - want).max() < 1e-7


def test_scalar_multistep_generator():
    """blur_jax/multistep.py on the scalar VP SDE: order-0 coefficient vs adaptive quadrature, row-sum identity,
    warm start, and the product's vectorised generator against the literal oracle."""
    import scipy.integrate
    from gddim_amd.blur import multistep
    sde = oblur.ScalarVPSDE()
    ts = np.linspace(0.9, 0.1, 6)
    c2 = oblur.ms_get_ab_eps_coef(sde, 3, ts, 2, num_item=2000)
    c0 = oblur.ms_get_ab_eps_coef(sde, 3, ts, 0, num_item=2000)
    assert c2.shape == (5, 4) and np.all(c2[:, 3] == 0) and np.all(c2[0, 1:] == 0) and c2[1, 2] == 0
    assert np.abs(c2.sum(axis=1) - c0[:, 0]).max() < 1e-12
    exact, _ = scipy.integrate.quad(lambda tau: sde.psi(tau, ts[1]) * sde.eps_integrand(tau), ts[0], ts[1])
    assert abs(c0[0, 0] - exact) / abs(exact) < 2e-3
    got = multistep.get_ab_eps_coef(sde, 3, ts, 2, num_item=2000)
    assert np.abs(got - c2).max() < 1e-12
    # Dirac data for the VP SDE: x_t = sqrt(alpha_t) c + sqrt(1-alpha_t) z, eps* = z: the AB loop is then exact
    z, c = 0.7, 0.2
    x = np.sqrt(sde.alpha(ts[0])) * c + np.sqrt(1 - sde.alpha(ts[0])) * z
    hist = np.zeros((3,))
    for i in range(5):
        coef = np.concatenate([[sde.psi(ts[i], ts[i + 1])], c2[i]])
        x, hist = oblur.ms_ab_step(np.array(x), coef, np.array(z), hist)
    assert abs(x - (np.sqrt(sde.alpha(ts[-1])) * c + np.sqrt(1 - sde.alpha(ts[-1])) * z)) < 5e-3


def test_product_blur_schedule_matches_oracle(native_lib):
    from gddim_amd.blur import sampling, sde_lib
    sde, osde = sde_lib.SDE(), oblur.SDE()
    rev = sampling.get_rev_ts(sde, 2, 50)
    assert np.abs(rev - oblur.get_rev_ts(osde, 2, 50)).max() < 1e-15
    assert np.abs(sde.y_mean_coef(rev) - osde.y_mean_coef(rev)).max() < 1e-15
    assert np.abs(sde.y_std_coef(rev) - osde.y_std_coef(rev)).max() < 1e-15


# ---- per-frequency multistep extension (gddim_amd/blur/multistep.get_ab_eps_coef_freq) -------------------------------------------
def test_freq_deis_coefficients():
    from gddim_amd.blur import multistep, sampling, sde_lib
    sde = sde_lib.SDE()
    osde = oblur.SDE()
    rev = sampling.get_rev_ts(sde, 2, 6)
    coef, psi = multistep.get_ab_eps_coef_freq(sde, 2, rev, 2)
    assert coef.shape == (6, 3, 32, 32) and psi.shape == (6, 32, 32)
    m, s = sde.y_mean_coef(rev)[..., 0], sde.y_std_coef(rev)
    order0 = s[1:, None, None] - psi * s[:-1, None, None]                 # the exact DDIM coefficient (blur_jax/sampling.py:69-74)
    assert np.abs(psi - m[1:] / m[:-1]).max() == 0
    # Lagrange weights sum to one: the slots of every row add up to the order-0 coefficient; row 0 is order 0 itself (warm start)
    scale = np.abs(order0).max()
    assert np.abs(coef.sum(axis=1) - order0).max() < 1e-12 * scale
    assert np.abs(coef[0, 0] - order0[0]).max() < 1e-12 * scale and np.abs(coef[0, 1:]).max() == 0 and np.abs(coef[1, 2]).max() == 0
    c0, _ = multistep.get_ab_eps_coef_freq(sde, 0, rev, 0)
    assert np.abs(c0[:, 0] - order0).max() < 1e-12 * scale
    # against adaptive quadrature of l_j dPhi/dtau (independent formulation in the oracle), a few rows
    want = oblur.freq_deis_coef(osde, rev[:4], 2)
    got = coef[:3]
    assert np.abs(got - want).max() < 2e-6 * np.abs(want).max()


def test_freq_deis_sampler_properties():
    """With eps_y(t) = a + b t per frequency (any linear-in-time prediction) the order-1 update integrates the eps term exactly, and
    with constant eps every order reproduces the exact trajectory y_t = m(t) y_0 + s(t) z (Dirac data)."""
    from gddim_amd.blur import multistep, sampling, sde_lib
    sde = sde_lib.SDE()
    osde = oblur.SDE()
    rng = np.random.default_rng(0)
    z = rng.standard_normal((2, 32, 32, 3))
    y0 = rng.standard_normal((2, 32, 32, 3))
    nfe = 7
    rev = sampling.get_rev_ts(sde, 2, nfe)
    m, s = osde.y_mean_coef(rev), osde.y_std_coef(rev)
    yT = m[0][None] * y0 + s[0] * z
    for order in (0, 1, 2, 3):
        coef, psi = multistep.get_ab_eps_coef_freq(sde, order, rev, order)
        x, _ = oblur.freq_deis_sampler(osde, lambda x_, t_: osde.y2x(z), yT, nfe, order, coef=coef, psi=psi)
        want = osde.y2x(m[-1][None] * y0 + s[-1] * z)
        assert np.abs(x - want).max() < 1e-9 * max(1.0, np.abs(want).max()), order
    # order of accuracy, step by step (whole trajectories are dominated by the order-0 warm-up step, the largest one of the
    # quadratic time grid): past the warm-up rows, an order-q row integrates the eps term exactly for any eps_y(t) that is a polynomial
    # of degree <= q in t when it is fed the exact history.  This pins the slot order (0 = newest) and the Lagrange nodes.
    z1, z2 = rng.standard_normal((32, 32)), rng.standard_normal((32, 32))
    z0 = rng.standard_normal((32, 32))
    for q in (1, 2):
        coef, psi = multistep.get_ab_eps_coef_freq(sde, q, rev, q, num_item=4000)
        poly = (lambda t: z0 + z1 * t) if q == 1 else (lambda t: z0 + z1 * t + z2 * t * t)
        for i in (q, q + 1, nfe - 1):
            got = sum(coef[i, o] * poly(rev[i - o]) for o in range(q + 1))
            grid = np.linspace(rev[i], rev[i + 1], 4001)
            mg, sg = osde.y_mean_coef(grid)[..., 0], osde.y_std_coef(grid)
            phi = mg[-1][None] / mg * sg[:, None, None]
            mid = 0.5 * (grid[1:] + grid[:-1])
            want = sum((phi[k + 1] - phi[k]) * poly(mid[k]) for k in range(len(mid)))
            assert np.abs(got - want).max() < 1e-6 * np.abs(want).max(), (q, i)
            # and the order-0 coefficient alone is NOT exact for it (the test has teeth)
            crude = (sg[-1] - psi[i] * sg[0]) * poly(rev[i])
            assert np.abs(crude - want).max() > 1e-3 * np.abs(want).max()
