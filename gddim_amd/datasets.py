"""Data (inverse) scalers -- the only part of the reference's datasets.py on the sampling path
(cld_jax/datasets.py:25-40, identical in blur_jax)."""


def get_data_scaler(config):
    """Data normalizer. Assume data are always in [0, 1]."""
    if config.data.centered:
        return lambda x: x * 2. - 1.
    return lambda x: x


def get_data_inverse_scaler(config):
    """Inverse data normalizer."""
    if config.data.centered:
        return lambda x: (x + 1.) / 2.
    return lambda x: x
