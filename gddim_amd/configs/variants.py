"""The other model configurations the reference ships, as thin edits of the two defaults.

Field values of cld_jax/configs/{deep,ndeep,ddpmpp,simple}_cifar10_config.py and blur_jax/configs/{ddpmpp,simple,debug}_cifar10_config.py
(the 2-D point config uses a toy MLP that is out of scope; the CelebA one differs from ddpmpp_cifar10 by image size only).  `get_config(name)` returns a ConfigDict with the same
fields as the reference's `get_config()` of that file:

  cld_deep_cifar10      accr architecture, reference-default R(t) scan (R_dt = 1e-5, Euler)
  cld_ndeep_cifar10     cld_deep + mixed_score = True, RK4 scan with R_dt = 1e-6
  cld_ddpmpp_cifar10    nf 128, 4 res blocks, positional embedding, fir = False, no input pyramid
  cld_simple_cifar10    the same with nf = 32
  cld_ddpmpp_celeba     cld_ddpmpp at 64x64 (CELEBA)
  blur_ddpmpp_cifar10 / blur_simple_cifar10 (nf 32) / blur_debug_cifar10 (nf 64)
"""
from . import blur_ddpm_deep_cifar10, blur_default_cifar10, cld_accr_dcifar10, cld_default_cifar10


def _ncsnpp_small(model, nf):
    cld_accr_dcifar10._ncsnpp_deep(model)
    model.nf = nf
    model.num_res_blocks = 4
    model.fir = False
    model.progressive_input = 'none'
    model.embedding_type = 'positional'


def _training(config, n_iters=None, eval_freq=2000, snapshot=10000):
    t = config.training
    if n_iters is not None:
        t.n_iters = n_iters
    t.continuous = True
    t.reduce_mean = True
    t.log_freq = 100
    t.eval_freq = eval_freq
    t.snapshot_freq_for_sampling = snapshot
    config.data.centered = True


def _cld_deep():
    config = cld_default_cifar10.get_default_configs()
    _training(config, n_iters=800001)
    cld_accr_dcifar10._ncsnpp_deep(config.model)
    return config


def _cld_ndeep():
    config = _cld_deep()
    config.model.mixed_score = True
    config.model.is_R_rk = True
    config.model.R_dt = 1e-6
    return config


def _cld_small(nf, eval_freq=2000, snapshot=10000):
    config = cld_default_cifar10.get_default_configs()
    _training(config, eval_freq=eval_freq, snapshot=snapshot)
    _ncsnpp_small(config.model, nf)
    return config


def _blur_small(nf, eval_freq=2000, snapshot=10000):
    config = blur_default_cifar10.get_default_configs()
    _training(config, eval_freq=eval_freq, snapshot=snapshot)
    _ncsnpp_small(config.model, nf)
    return config


def _cld_celeba():
    # cld_jax/configs/ddpmpp_celeba_config.py: a self-contained file in the reference; only what differs from cld_ddpmpp_cifar10
    config = _cld_small(128)
    config.data.dataset = 'CELEBA'
    config.data.image_size = 64
    config.model.ema_rate = 0.999
    return config


_TABLE = {
    'cld_accr_dcifar10': cld_accr_dcifar10.get_config,
    'cld_deep_cifar10': _cld_deep,
    'cld_ndeep_cifar10': _cld_ndeep,
    'cld_ddpmpp_cifar10': lambda: _cld_small(128),
    'cld_simple_cifar10': lambda: _cld_small(32, eval_freq=500, snapshot=3000),
    'cld_ddpmpp_celeba': _cld_celeba,
    'blur_ddpm_deep_cifar10': blur_ddpm_deep_cifar10.get_config,
    'blur_ddpmpp_cifar10': lambda: _blur_small(128),
    'blur_simple_cifar10': lambda: _blur_small(32, eval_freq=500, snapshot=3000),
    'blur_debug_cifar10': lambda: _blur_small(64),
}


def names():
    return sorted(_TABLE)


def get_config(name):
    if name not in _TABLE:
        raise KeyError("unknown configuration %r (have: %s)" % (name, ", ".join(names())))
    return _TABLE[name]()
