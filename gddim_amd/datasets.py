"""The two data scalers the samplers receive (cld_jax/datasets.py:25-40, identical in blur_jax): images live in [0, 1], the
models in [-1, 1] when config.data.centered."""


def _affine(scale, shift):
    def fn(x):
        return x * scale + shift
    return fn


def get_data_scaler(config):
    return _affine(2.0, -1.0) if config.data.centered else _affine(1.0, 0.0)


def get_data_inverse_scaler(config):
    return _affine(0.5, 0.5) if config.data.centered else _affine(1.0, 0.0)
