"""cld_jax/configs/default_cifar10_config.py:4-101 (`get_default_configs`): field names and defaults, from configs/_base.py."""
from . import _base


def get_default_configs():
    return _base.make(_base.COMMON, _base.CLD)
