"""gddim_amd -- MI355X-native gDDIM sampling engine (the sampler hot path of qsh-zh/gDDIM).

Layout mirrors the reference's two packages for this path only:
  gddim_amd.cld.{sampling,sde_lib,deis,utils}      <- cld_jax/{sampling,sde_lib,deis,utils}.py
  gddim_amd.blur.{sampling,sde_lib,blur,multistep} <- blur_jax/{sampling,sde_lib,blur,multistep}.py
  gddim_amd.models.{ncsnpp,utils}                  <- */models/{ncsnpp,utils}.py
  gddim_amd.configs, gddim_amd.datasets            <- config field names, (inverse) scalers
  gddim_amd.engine / ops / _abi                    <- ctypes binding of include/gddim_amd.h
  gddim_amd.csrc                                   <- HIP kernels (gfx950) + C ABI
"""
__version__ = "0.1.0"
