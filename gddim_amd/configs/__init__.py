"""Config trees with the reference's field names and defaults (the sampler-relevant API).

  cld_default_cifar10   <- cld_jax/configs/default_cifar10_config.py
  cld_accr_dcifar10     <- cld_jax/configs/accr_dcifar10_config.py   (the benchmarked CLD config)
  blur_default_cifar10  <- blur_jax/configs/default_cifar10_config.py
  blur_ddpm_deep_cifar10<- blur_jax/configs/ddpm_deep_cifar10_config.py
"""
from . import cld_default_cifar10, cld_accr_dcifar10, blur_default_cifar10, blur_ddpm_deep_cifar10  # noqa: F401
