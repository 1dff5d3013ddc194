"""Reader (and writer, for round-trip tests and exports) of the reference's checkpoint files.

The reference saves its `State` dataclass with `flax.training.checkpoints.save_checkpoint(dir, state, step, ...)`
(cld_jax/run_lib.py:197,240,486-630) and reads it back with `restore_checkpoint(dir_or_file, state, step=None)`
(run_lib.py:112,423-430,707).  flax (0.3.1, not installed here) writes one file `<prefix><step>` holding
`flax.serialization.to_bytes(state)`: a msgpack map of the dataclass' state dict
    {'step', 'optimizer': {'target': params, 'state': ...}, 'lr', 'model_state', 'ema_rate', 'params_ema': params, 'rng'}
in which every array is a msgpack ExtType:
    code 1  ndarray       payload = msgpack((shape, dtype.name, arr.tobytes('C')))
    code 2  complex       payload = msgpack((real, imag))
    code 3  numpy scalar  payload as code 1 with shape ()
and arrays above 2**30 bytes are split into {'__msgpack_chunked_array__': True, 'shape': ..., 'chunks': {'0': ..}}.
This module restates that published format with the `msgpack` package; it needs neither jax nor flax.  The sampling path
only consumes `params_ema` (the nested dict of the flax module tree, whose names gddim_amd.engine.Engine.load_params expects)
and `model_state`.
"""
import os
import re

import numpy as np

from .models.utils import State

_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3
_CHUNK_KEY = '__msgpack_chunked_array__'
MAX_CHUNK_SIZE = 2 ** 30


def _msgpack():
    import msgpack
    return msgpack


def _ndarray_from_bytes(data):
    shape, dtype_name, buf = _msgpack().unpackb(data, raw=True)
    name = dtype_name.decode() if isinstance(dtype_name, bytes) else dtype_name
    if name == 'bfloat16':  # stored as raw 16-bit words; widen to float32
        raw = np.frombuffer(buf, dtype=np.uint16).astype(np.uint32) << 16
        return raw.view(np.float32).reshape(shape)
    return np.frombuffer(buf, dtype=np.dtype(name)).reshape(shape)


def _ext_hook(code, data):
    if code == _EXT_NDARRAY:
        return _ndarray_from_bytes(data)
    if code == _EXT_COMPLEX:
        re_, im_ = _msgpack().unpackb(data)
        return complex(re_, im_)
    if code == _EXT_NPSCALAR:
        return _ndarray_from_bytes(data)[()]
    return _msgpack().ExtType(code, data)


def _unchunk(tree):
    if isinstance(tree, dict):
        if tree.get(_CHUNK_KEY, False):
            shape = tuple(tree['shape'])
            chunks = tree['chunks']
            flat = np.concatenate([np.asarray(chunks[str(i)]).ravel() for i in range(len(chunks))])
            return flat.reshape(shape)
        return {k: _unchunk(v) for k, v in tree.items()}
    return tree


def msgpack_restore(encoded):
    """flax.serialization.msgpack_restore: bytes -> nested dict of numpy arrays / python scalars."""
    tree = _msgpack().unpackb(encoded, ext_hook=_ext_hook, raw=False, strict_map_key=False)
    return _unchunk(tree)


def _pack_array(a):
    a = np.asarray(a, order='C')  # (ascontiguousarray would promote 0-d arrays to shape (1,))
    return _msgpack().ExtType(_EXT_NDARRAY, _msgpack().packb((a.shape, a.dtype.name, a.tobytes('C')), use_bin_type=True))


def _chunk(tree):
    if isinstance(tree, dict):
        return {k: _chunk(v) for k, v in tree.items()}
    if isinstance(tree, np.ndarray) and tree.nbytes > MAX_CHUNK_SIZE:
        flat = tree.ravel()
        n = max(1, MAX_CHUNK_SIZE // tree.dtype.itemsize)
        return {_CHUNK_KEY: True, 'shape': list(tree.shape),
                'chunks': {str(i): flat[o:o + n] for i, o in enumerate(range(0, flat.size, n))}}
    return tree


def msgpack_serialize(tree):
    """flax.serialization.msgpack_serialize: nested dict of arrays / scalars -> bytes."""
    def default(o):
        if isinstance(o, np.ndarray):
            return _pack_array(o)
        if isinstance(o, np.generic):
            return _msgpack().ExtType(_EXT_NPSCALAR, _pack_array(np.asarray(o)).data)
        if isinstance(o, complex):
            return _msgpack().ExtType(_EXT_COMPLEX, _msgpack().packb((o.real, o.imag)))
        if hasattr(o, 'detach'):
            return _pack_array(o.detach().cpu().numpy())
        raise TypeError("cannot serialise %r" % type(o))
    return _msgpack().packb(_chunk(tree), default=default, strict_types=True, use_bin_type=True)


def _natural_key(path):
    return [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', path)]


def latest_checkpoint(ckpt_dir, prefix='checkpoint_'):
    """flax.training.checkpoints.latest_checkpoint: the file `<prefix><step>` with the largest step (natural order)."""
    if not os.path.isdir(ckpt_dir):
        return None
    files = [f for f in os.listdir(ckpt_dir) if f.startswith(prefix) and not f.endswith('tmp')]
    return os.path.join(ckpt_dir, sorted(files, key=_natural_key)[-1]) if files else None


def restore_checkpoint(ckpt_dir, target=None, step=None, prefix='checkpoint_'):
    """flax.training.checkpoints.restore_checkpoint as the reference calls it.

    ckpt_dir: a checkpoint file, or a directory (then `step` selects `<prefix><step>`, default the latest).
    target:   a `State` (or None).  Returns a State whose fields come from the file; when nothing is found the target is
              returned unchanged, like flax does."""
    if os.path.isdir(ckpt_dir):
        path = os.path.join(ckpt_dir, '%s%s' % (prefix, step)) if step is not None else latest_checkpoint(ckpt_dir, prefix)
        if path is None or not os.path.exists(path):
            if step is not None:
                raise ValueError("Matching checkpoint not found: %s" % path)
            return target
    else:
        path = ckpt_dir
        if not os.path.exists(path):
            return target
    with open(path, 'rb') as f:
        tree = msgpack_restore(f.read())
    return state_from_dict(tree, target)


def state_from_dict(tree, target=None):
    base = target if isinstance(target, State) else State()
    opt = tree.get('optimizer', base.optimizer)
    return base.replace(step=int(np.asarray(tree.get('step', base.step))), optimizer=opt,
                        lr=float(np.asarray(tree.get('lr', base.lr))), model_state=tree.get('model_state') or {},
                        ema_rate=float(np.asarray(tree.get('ema_rate', base.ema_rate))),
                        params_ema=tree.get('params_ema', base.params_ema), rng=tree.get('rng', base.rng))


def state_to_dict(state):
    return {'step': np.asarray(state.step), 'optimizer': state.optimizer if state.optimizer is not None else {},
            'lr': np.asarray(state.lr, dtype=np.float32), 'model_state': state.model_state or {},
            'ema_rate': np.asarray(state.ema_rate, dtype=np.float32), 'params_ema': state.params_ema,
            'rng': state.rng if state.rng is not None else np.zeros(2, dtype=np.uint32)}


def save_checkpoint(ckpt_dir, target, step, prefix='checkpoint_', keep=1, overwrite=False):
    """Write `<ckpt_dir>/<prefix><step>` in the same format (used to export synthetic weights and by the tests)."""
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, '%s%s' % (prefix, step))
    if os.path.exists(path) and not overwrite:
        raise ValueError("checkpoint %s exists" % path)
    tree = state_to_dict(target) if isinstance(target, State) else target
    tmp = path + 'tmp'
    with open(tmp, 'wb') as f:
        f.write(msgpack_serialize(tree))
    os.replace(tmp, path)
    files = sorted([f for f in os.listdir(ckpt_dir) if f.startswith(prefix) and not f.endswith('tmp')], key=_natural_key)
    for old in files[:-keep] if keep and len(files) > keep else []:
        os.remove(os.path.join(ckpt_dir, old))
    return path
