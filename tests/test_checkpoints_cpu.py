"""The flax msgpack checkpoint format (gddim_amd/checkpoints.py) without flax: byte-level known answers of the published
encoding, round trips, chunked arrays, directory handling, and that a restored tree is what Engine.load_params expects."""
import os

import msgpack
import numpy as np
import pytest

from conftest import tiny_config
from gddim_amd import checkpoints as ck
from gddim_amd.models.utils import State
from oracle import ncsnpp as onet


def test_ndarray_ext_encoding_is_the_published_one():
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    enc = ck.msgpack_serialize({'w': a})
    # outer map {'w': ExtType(1, msgpack((shape, dtype name, C-order bytes)))}
    raw = msgpack.unpackb(enc, raw=False)
    assert list(raw) == ['w'] and isinstance(raw['w'], msgpack.ExtType) and raw['w'].code == 1
    shape, name, buf = msgpack.unpackb(raw['w'].data, raw=False)
    assert tuple(shape) == (2, 3) and name == 'float32' and buf == a.tobytes('C')
    # hand-built payload (what flax's _ndarray_to_bytes produces) decodes
    hand = msgpack.packb({'k': msgpack.ExtType(1, msgpack.packb(((3,), 'int32', np.array([7, 8, 9], np.int32).tobytes()), use_bin_type=True)),
                          's': msgpack.ExtType(3, msgpack.packb(((), 'float64', np.float64(2.5).tobytes()), use_bin_type=True)),
                          'c': msgpack.ExtType(2, msgpack.packb((1.0, -2.0)))}, use_bin_type=True)
    out = ck.msgpack_restore(hand)
    assert out['k'].tolist() == [7, 8, 9] and out['k'].dtype == np.int32 and out['s'] == 2.5 and out['c'] == complex(1, -2)


def test_chunked_arrays_round_trip(monkeypatch):
    monkeypatch.setattr(ck, 'MAX_CHUNK_SIZE', 64)
    a = np.random.default_rng(0).standard_normal((5, 7)).astype(np.float32)  # 140 bytes -> 3 chunks
    enc = ck.msgpack_serialize({'a': a, 'b': {'c': np.int64(3)}})
    raw = msgpack.unpackb(enc, raw=False, ext_hook=lambda c, d: ('ext', c))
    assert raw['a']['__msgpack_chunked_array__'] is True and raw['a']['shape'] == [5, 7] and len(raw['a']['chunks']) == 3
    out = ck.msgpack_restore(enc)
    assert np.array_equal(out['a'], a) and out['b']['c'] == 3


def test_state_round_trip_and_directory_semantics(tmp_path):
    cfg = tiny_config("cld", nf=16, num_res_blocks=1, image_size=8, ch_mult=(1, 2), attn=(4,))
    params = onet.random_params(cfg, 6, seed=1)
    tree = onet.unflatten(params) if hasattr(onet, 'unflatten') and not any(isinstance(v, dict) for v in params.values()) else params
    st = State(step=7, lr=2e-4, ema_rate=0.9999, params_ema=tree, rng=np.array([1, 2], np.uint32))
    d = str(tmp_path / "ckpt")
    p3 = ck.save_checkpoint(d, st.replace(step=3), 3, keep=5)
    p12 = ck.save_checkpoint(d, st.replace(step=12), 12, keep=5)
    assert os.path.basename(p12) == "checkpoint_12" and ck.latest_checkpoint(d) == p12  # natural order: 12 after 3
    back = ck.restore_checkpoint(d, State())
    assert back.step == 12 and abs(back.ema_rate - 0.9999) < 1e-7 and back.rng.tolist() == [1, 2]
    assert ck.restore_checkpoint(d, State(), step=3).step == 3 and ck.restore_checkpoint(p3, None).step == 3
    with pytest.raises(ValueError):
        ck.restore_checkpoint(d, State(), step=4)
    untouched = State(step=99)
    assert ck.restore_checkpoint(str(tmp_path / "missing"), untouched) is untouched

    def flat(t, pre=""):
        out = {}
        for k, v in t.items():
            out.update(flat(v, pre + k + "/") if isinstance(v, dict) else {pre + k: v})
        return out

    fa, fb = flat(tree), flat(back.params_ema)
    assert fa.keys() == fb.keys() and all(np.array_equal(np.asarray(fa[k]), fb[k]) for k in fa)


def test_restored_tree_loads_into_a_plan_only_engine(native_lib, tmp_path):
    from gddim_amd.engine import Engine
    cfg = tiny_config("cld", nf=16, num_res_blocks=1, image_size=8, ch_mult=(1, 2), attn=(4,))
    params = onet.random_params(cfg, 6, seed=3)
    ck.save_checkpoint(str(tmp_path), State(params_ema=params), 1)
    st = ck.restore_checkpoint(str(tmp_path), State())
    eng = Engine(cfg, 6, 2, plan_only=True)
    eng.load_params(st.params_ema)  # names and shapes are checked natively; raises on any mismatch
