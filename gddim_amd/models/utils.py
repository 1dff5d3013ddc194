"""Model helper API of the reference (cld_jax/models/utils.py, blur_jax/models/utils.py) on the native engine.

  State                      cld_jax/models/utils.py:32-40
  register_model/get_model   :43-66
  init_model                 :109-125   (blur: blur_jax/models/utils.py:85-101, no velocity channels)
  get_model_fn               :128-166   'b ... d g -> b ... (g d)' packing for CLD
  get_eps_fn                 :168-182   labels = 999 t (+ the mixed_score term)
  get_score_fn               :184-211
  get_yeps_fn                blur_jax/models/utils.py:155-160
"""
import numpy as np

from .. import ops
from .._abi import GddimError

_MODELS = {}


class State(object):
    """The fields of the reference's flax dataclass; sampling reads params_ema and model_state only."""

    def __init__(self, step=0, optimizer=None, lr=0.0, model_state=None, ema_rate=0.0, params_ema=None, rng=None):
        self.step = step
        self.optimizer = optimizer
        self.lr = lr
        self.model_state = {} if model_state is None else model_state
        self.ema_rate = ema_rate
        self.params_ema = params_ema
        self.rng = rng

    def replace(self, **kw):
        d = dict(self.__dict__)
        d.update(kw)
        return State(**d)


def register_model(cls=None, *, name=None):
    def _register(c):
        local_name = c.__name__ if name is None else name
        if local_name in _MODELS:
            raise ValueError('Already registered model with name: %s' % local_name)
        _MODELS[local_name] = c
        return c
    return _register if cls is None else _register(cls)


def get_model(name):
    from . import ncsnpp  # noqa: F401  (registers 'ncsnpp')
    return _MODELS[name]


def as_numpy_rng(rng):
    """Accept an int seed, a numpy Generator or a (stack of) key-like array(s); jax PRNG keys cannot be
    reproduced without jax, so any of these only seeds a numpy Generator."""
    if isinstance(rng, np.random.Generator):
        return rng
    if hasattr(rng, 'detach'):
        rng = rng.detach().cpu().numpy()
    a = np.asarray(rng).astype(np.uint64).ravel()
    return np.random.default_rng(a.tolist() if a.size > 1 else int(a[0]))


def unflatten(flat):
    tree = {}
    for name, a in flat.items():
        parts = name.split('/')
        d = tree
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = a
    return tree


def _is_cld(config):
    return hasattr(config.model, 'm_inv')


def init_model(rng, config, init_scale_override=None):
    """Initialize a model: returns (model, init_model_state, initial_params)."""
    from . import ncsnpp
    model = get_model(config.model.name)(config=config)
    in_ch = config.data.num_channels * (2 if _is_cld(config) else 1)
    params = ncsnpp.init_params(config, in_ch, as_numpy_rng(rng), init_scale_override)
    return model, {}, params


def get_model_fn(model, params, states, train=False, packed_pairs=True, slot=0):
    """model_fn(x, labels, rng=None) -> (output, states).  packed_pairs: x is a CLD state (B,...,d,2)."""

    def model_fn(x, labels, rng=None):
        if train:
            raise GddimError("training is out of scope of the gddim_amd sampling engine")
        variables = {'params': params, **states}
        if packed_pairs:
            eng = model.engine_for(params, 2 * x.shape[-2], x.shape[0], x.device.index, slot=slot)
            # gd_cld_eps takes t and multiplies by 999 itself; labels come in as 999 t
            return eng.cld_eps(x, labels / 999.0), states
        return model.apply(variables, x, labels, train=False, mutable=False), states

    return model_fn


def get_eps_fn(sde, model, params, states, train=False, continuous=False, return_state=False, slot=0):
    """slot: which of this process's engines evaluates (gddim_amd.devices: one engine per local device of a psampler call)."""
    cld = hasattr(sde, 'mixed_score')

    def eps_fn(x, t, rng=None):
        if train:
            raise GddimError("training is out of scope of the gddim_amd sampling engine")
        if cld:
            eng = model.engine_for(params, 2 * x.shape[-2], x.shape[0], x.device.index, slot=slot)
            out = eng.cld_eps(x, t)  # labels = t * 999 inside (models/utils.py:172)
            if sde.mixed_score:  # + bmm(v_invR(t), x.at[..., 0].set(0)) (models/utils.py:174-176)
                from .. import ops
                tt = t.detach().cpu().numpy() if hasattr(t, 'detach') else np.asarray(t)
                m = np.array(sde.v_invR(tt), dtype=np.float32)
                m[:, :, 0] = 0.0
                out = out + ops.bmm(m, x)
        else:
            labels = sde.encode_t(t)
            out = model.apply({'params': params, **states}, sde.encode_x(x), labels, train=False, mutable=False)
            out = sde.model2eps(x, t, out)
        return (out, states) if return_state else out

    return eps_fn


def get_score_fn(sde, model, params, states, train=False, continuous=False, return_state=False, slot=0):
    eps_fn = get_eps_fn(sde, model, params, states, train, continuous, return_state, slot=slot)

    def score_fn(x, t, rng=None):
        out = eps_fn(x, t, rng)
        if return_state:
            out, state = out
        score = sde.eps2score(out, t)
        return (score, state) if return_state else score

    return score_fn


def get_yeps_fn(sde, model, params, states, train=False, continuous=False):
    xeps_fn = get_eps_fn(sde, model, params, states, train, continuous, return_state=False)

    def eps_fn(y, t, rng=None):
        return sde.x2y(xeps_fn(sde.y2x(y), t, rng))

    return eps_fn
