"""cld_jax/configs/accr_dcifar10_config.py:4-49, the CLD gDDIM benchmark config: the deep NCSN++ and the accurate R(t) scan."""
from . import _base
from .cld_default_cifar10 import get_default_configs


def _ncsnpp_deep(model):
    for field, value in _base.NCSNPP_DEEP.items():
        setattr(model, field, list(value) if isinstance(value, list) else value)


def _long_run(training, n_iters):
    for field, value in dict(n_iters=n_iters, continuous=True, reduce_mean=True, log_freq=100, eval_freq=2000,
                             snapshot_freq_for_sampling=10000).items():
        setattr(training, field, value)


def get_config():
    config = get_default_configs()
    _long_run(config.training, 1000001)
    config.model.mixed_score, config.model.is_R_rk, config.model.R_dt = False, True, 1e-6
    config.data.centered = True
    _ncsnpp_deep(config.model)
    return config
