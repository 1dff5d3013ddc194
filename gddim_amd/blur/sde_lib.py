"""Blurring-diffusion SDE: the members of blur_jax/sde_lib.py:18-154 that the sampler touches, on the host in float64.

Schedule (32 x 32 DCT grid hard-wired by the reference, :24-26):
    alpha(t)      = cos^2(pi/2 * (t + 0.004) / 1.008)
    sigma_B(t)    = sigma_blur_max * sin^2(pi t / 2)
    D_t[h, w]     = min_scale + (1 - min_scale) * exp(-sigma_B(t)^2 / 2 * ((pi h / 32)^2 + (pi w / 32)^2))
    mean / std    : y_t = sqrt(alpha) D_t y_0 + sqrt(1 - alpha) eps            (y = DCT(x))
    sampling_T    = t at which sqrt(1 - alpha) / sqrt(alpha) reaches rho = 80 measured from alpha(0)  (rho2t)
Public names (SDE, from_config and the method names) are the reference's, the bodies are vectorised numpy."""
import numpy as np

from . import blur

_GRID = 32
_HALF_PI = np.pi / 2


def _f64(t):
    return np.atleast_1d(np.asarray(t, dtype=np.float64))


class SDE(object):
    T = 1.0

    def __init__(self, min_scale=0.001, sigma_blur_max=10.0, sampling_eps=1e-5):
        self.min_scale, self.sigma_blur_max, self.sampling_eps = min_scale, sigma_blur_max, sampling_eps
        w2 = (np.pi * np.arange(_GRID) / _GRID) ** 2                  # squared DCT frequencies along one axis
        self.labda = (w2[:, None] + w2[None, :])[None, :, :, None]    # (1, 32, 32, 1): lambda_hw of the heat kernel
        self.alpha_start = self.t2alpha_fn(0.0)

    # ---- noise schedule ----
    def t2alpha_fn(self, t):
        return np.square(np.cos(_HALF_PI * (np.asarray(t, dtype=np.float64) + 0.004) / 1.008))

    def alpha2t_fn(self, alpha):
        return 1.008 * np.arccos(np.sqrt(alpha)) / _HALF_PI - 0.004

    def rho2t(self, rho):
        a0 = self.alpha_start
        return self.alpha2t_fn(a0 / (a0 + np.square(rho + np.sqrt(1 - a0))))

    @property
    def sampling_T(self):
        return self.rho2t(80.0)

    # ---- blur schedule ----
    def get_frequency_scaling(self, t):
        tau = 0.5 * np.square(self.sigma_blur_max * np.square(np.sin(_HALF_PI * _f64(t))))  # dissipation time per sample
        return self.min_scale + (1 - self.min_scale) * np.exp(-tau[:, None, None, None] * self.labda)

    def y_mean_coef(self, ts):
        ts = _f64(ts)
        return self.get_frequency_scaling(ts) * np.sqrt(self.t2alpha_fn(ts))[:, None, None, None]  # (B, 32, 32, 1)

    def y_std_coef(self, ts):
        return np.sqrt(1 - self.t2alpha_fn(_f64(ts)))

    # ---- device-side members ----
    def prior_sampling(self, rng, shape, device=None):
        """N(0, I) in DCT space (:134-136); drawn with torch's device generator (jax's threefry stream is not reproducible)."""
        import torch
        device = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        gen = torch.Generator(device=device)
        gen.manual_seed(int(rng.integers(0, 2 ** 62)) if hasattr(rng, 'integers') else int(rng))
        return torch.randn(tuple(shape), generator=gen, device=device, dtype=torch.float32)

    x2y = staticmethod(blur.batch_img_dct)
    y2x = staticmethod(blur.batch_img_idct)

    @staticmethod
    def encode_t(t):
        return t * 999

    @staticmethod
    def encode_x(xs):
        return xs

    @staticmethod
    def model2eps(xs, ts, model_output):
        return model_output


def from_config(config):
    return SDE(sigma_blur_max=config.model.sigma_blur_max, sampling_eps=config.sampling.t0)
