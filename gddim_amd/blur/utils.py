"""blur_jax/utils.py: get_data_shape (:172-177 of the CLD twin)."""


def get_data_shape(config):
    if "ps" in config.data.dataset:
        return (config.data.dim,)
    return (config.data.image_size, config.data.image_size, config.data.num_channels)
