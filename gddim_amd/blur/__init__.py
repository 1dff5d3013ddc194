"""Blurring-diffusion gDDIM: the sampler hot path of the reference's blur_jax package on the native engine."""
