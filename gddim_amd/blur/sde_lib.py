"""Blurring-diffusion SDE: sampling members of blur_jax/sde_lib.py:18-154 (host side, float64 numpy).

alpha(t) = cos^2((t+.004)/1.008 * pi/2); D_t[h,w] = exp(-(sigma_max sin^2(pi t/2))^2/2 * lambda_hw)(1-min_scale)+min_scale,
lambda_hw = (pi h/32)^2 + (pi w/32)^2 (hard-wired 32x32 grid, :24-26); y_mean_coef = sqrt(alpha) D_t,
y_std_coef = sqrt(1-alpha); sampling starts at rho2t(80)."""
import numpy as np

from . import blur


class SDE(object):
    def __init__(self, min_scale=0.001, sigma_blur_max=10.0, sampling_eps=1e-5):
        self.min_scale = min_scale
        self.sigma_blur_max = sigma_blur_max
        self.sampling_eps = sampling_eps
        img_dim = 32
        freqs = np.pi * np.linspace(0, img_dim - 1, img_dim) / img_dim
        self.labda = freqs[None, :, None, None] ** 2 + freqs[None, None, :, None] ** 2
        self.alpha_start = self.t2alpha_fn(0.0)

    @property
    def T(self):
        return 1.0

    @property
    def sampling_T(self):
        return self.rho2t(80.0)

    def t2alpha_fn(self, t):
        return np.cos((np.asarray(t, dtype=np.float64) + 0.004) / 1.008 * np.pi / 2) ** 2

    def alpha2t_fn(self, alpha):
        return np.arccos(np.sqrt(alpha)) * 2 / np.pi * 1.008 - 0.004

    def rho2t(self, rho):
        num = self.alpha_start
        denum = (rho + np.sqrt(1 - self.alpha_start)) ** 2 + self.alpha_start
        return self.alpha2t_fn(num / denum)

    def get_frequency_scaling(self, t):
        t = np.atleast_1d(np.asarray(t, dtype=np.float64))
        sigma_blur = self.sigma_blur_max * np.sin(t * np.pi / 2) ** 2
        dissipation_time = sigma_blur ** 2 / 2
        logits = dissipation_time[:, None, None, None] * self.labda
        return np.exp(-logits) * (1 - self.min_scale) + self.min_scale

    def y_mean_coef(self, ts):
        ts = np.atleast_1d(np.asarray(ts, dtype=np.float64))
        return np.sqrt(self.t2alpha_fn(ts))[:, None, None, None] * self.get_frequency_scaling(ts)  # (B, 32, 32, 1)

    def y_std_coef(self, ts):
        return np.sqrt(1 - self.t2alpha_fn(np.atleast_1d(np.asarray(ts, dtype=np.float64))))

    def prior_sampling(self, rng, shape, device=None):
        import torch
        device = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        g = torch.Generator(device=device)
        g.manual_seed(int(rng) if not hasattr(rng, 'integers') else int(rng.integers(0, 2 ** 62)))
        return torch.randn(tuple(shape), generator=g, device=device, dtype=torch.float32)

    def x2y(self, xs):
        return blur.batch_img_dct(xs)

    def y2x(self, ys):
        return blur.batch_img_idct(ys)

    def encode_t(self, t):
        return 999 * t

    def encode_x(self, xs):
        return xs

    def model2eps(self, xs, ts, model_output):
        del xs, ts
        return model_output


def from_config(config):
    return SDE(sigma_blur_max=config.model.sigma_blur_max, sampling_eps=config.sampling.t0)
