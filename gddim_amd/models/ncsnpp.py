"""NCSN++ score network, engine backed.

Mirrors the call surface of the reference's flax module (cld_jax/models/ncsnpp.py:35-243):
`NCSNpp(config=config)` is stateless; `model.apply(variables, x, labels, train=False, mutable=False)`
evaluates it.  Here `apply` binds `variables['params']` to a native engine (cached per parameter
tree and batch capacity) and runs the HIP forward.  Training-mode apply is out of scope.
"""
import math

import numpy as np

from .. import engine as _engine
from .._abi import GddimError
from . import utils as mutils


@mutils.register_model(name='ncsnpp')
class NCSNpp(object):
    """NCSN++ model"""

    def __init__(self, config):
        self.config = config
        self._engines = {}

    # -- engine cache: one engine per (params identity, in_channels), grown when a larger batch arrives
    def engine_for(self, params, in_channels, batch, device=None):
        import torch
        dev = torch.cuda.current_device() if device is None else device
        key = (id(params), int(in_channels), dev)
        ent = self._engines.get(key)
        if ent is None or ent[1] < batch:
            if ent is not None:
                ent[0].close()
            eng = _engine.Engine(self.config, in_channels, max_batch=batch, device=dev)
            eng.load_params(params)
            ent = (eng, batch, params)  # keep `params` alive so id() stays unique
            self._engines[key] = ent
        return ent[0]

    def apply(self, variables, x, labels, train=False, mutable=False, rngs=None):
        if train:
            raise GddimError("training-mode apply is out of scope of the gddim_amd sampling engine")
        params = variables['params']
        eng = self.engine_for(params, x.shape[-1], x.shape[0], x.device.index)
        return eng.unet(x, labels)

    def init(self, rngs, fake_input, fake_label):
        """flax `model.init`: returns {'params': tree}.  rngs: {'params': seed-like}."""
        seed = rngs['params'] if isinstance(rngs, dict) else rngs
        return {'params': init_params(self.config, int(fake_input.shape[-1]), mutils.as_numpy_rng(seed))}


def param_shapes(config, in_channels):
    """Names/shapes of the flax parameter tree, straight from the native plan."""
    eng = _engine.Engine(config, in_channels, max_batch=1, plan_only=True)
    try:
        return eng.param_specs()
    finally:
        eng.close()


def init_params(config, in_channels, rng, init_scale_override=None):
    """Random parameters with the reference's initialisers:
    conv / dense / NIN kernels  variance_scaling(scale, 'fan_avg', 'uniform') (layers.py:60-63), where scale is
                                1 except init_scale (config.model.init_scale, 0 -> 1e-10) for every block's
                                Conv_1, attention NIN_3 and the head conv, and 0.1 for NIN_0..2 (layers.py:469);
    biases zeros; GroupNorm scale ones / bias zeros; Fourier W ~ N(0, fourier_scale^2) (layerspp.py:40).
    `init_scale_override` replaces config.model.init_scale (the flax default 0 makes eps ~ 0)."""
    m = config.model
    init_scale = m.init_scale if init_scale_override is None else init_scale_override
    flat = {}
    for name, shp in param_shapes(config, in_channels).items():
        parts = name.split('/')
        leaf = parts[-1]
        if name == 'GaussianFourierProjection_0/W':
            a = rng.standard_normal(shp) * m.fourier_scale
        elif leaf in ('kernel', 'weight', 'W'):
            scale = 1.0
            mod = parts[-2]
            if mod == 'Conv_1':  # second conv of a ResBlock, or the head conv
                scale = init_scale
            elif mod == 'NIN_3':
                scale = init_scale
            elif mod.startswith('NIN_'):
                scale = 0.1
            scale = 1e-10 if scale == 0 else scale
            rf = int(np.prod(shp[:-2])) if len(shp) > 2 else 1
            fan_in, fan_out = shp[-2] * rf, shp[-1] * rf
            lim = math.sqrt(3.0 * scale / ((fan_in + fan_out) / 2.0))
            a = (rng.random(shp, dtype=np.float32) * 2.0 - 1.0) * np.float32(lim)
        elif leaf == 'scale':
            a = np.ones(shp, dtype=np.float32)
        else:
            a = np.zeros(shp, dtype=np.float32)
        flat[name] = np.asarray(a, dtype=np.float32)
    return mutils.unflatten(flat)
