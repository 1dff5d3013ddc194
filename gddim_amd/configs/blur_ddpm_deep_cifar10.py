"""blur_jax/configs/ddpm_deep_cifar10_config.py:4-48, the blur gDDIM config: the same deep NCSN++ on 3 channels."""
from .blur_default_cifar10 import get_default_configs
from .cld_accr_dcifar10 import _long_run, _ncsnpp_deep


def get_config():
    config = get_default_configs()
    _long_run(config.training, 1000001)
    data = config.data
    data.centered, data.num_channels, data.is_partial, data.random_flip = True, 3, False, True
    _ncsnpp_deep(config.model)
    return config
