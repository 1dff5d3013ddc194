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
