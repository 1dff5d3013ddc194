"""Field names / values of blur_jax/configs/ddpm_deep_cifar10_config.py:4-48 (the blur gDDIM config)."""
from .blur_default_cifar10 import get_default_configs
from .cld_accr_dcifar10 import _ncsnpp_deep


def get_config():
    config = get_default_configs()
    training = config.training
    training.n_iters = 1000001
    training.continuous = True
    training.reduce_mean = True
    training.log_freq = 100
    training.eval_freq = 2000
    training.snapshot_freq_for_sampling = 10000
    data = config.data
    data.centered = True
    data.num_channels = 3
    data.is_partial = False
    data.random_flip = True
    _ncsnpp_deep(config.model)
    return config
