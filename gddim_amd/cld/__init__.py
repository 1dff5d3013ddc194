"""CLD gDDIM: the sampler hot path of the reference's cld_jax package on the native engine."""
