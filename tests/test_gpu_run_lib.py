"""Sampling rounds on the GPU (gddim_amd/run_lib.py: sample_data of cld_jax/run_lib.py:674-731): file names, keys, dtypes,
the uint8 quantisation, resume semantics and the checkpoint path."""
import os

import numpy as np
import pytest

from conftest import tiny_config
from oracle import ncsnpp as onet

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["cld", "blur"])
def test_sample_data_rounds(native_lib, tmp_path, kind):
    from gddim_amd import checkpoints, run_lib
    from gddim_amd.models.utils import State
    cfg = tiny_config(kind, nf=16, num_res_blocks=1, image_size=16 if kind == "cld" else 32, ch_mult=(1, 2), attn=(8,))
    cfg.eval.num_samples = 6
    cfg.eval.batch_size = 4
    cfg.sampling.nfe = 4
    if kind == "cld":
        cfg.sampling.method, cfg.sampling.deis_order = 'deis', 1
    params = onet.random_params(cfg, 6 if kind == "cld" else 3, seed=4)
    ckpt = checkpoints.save_checkpoint(str(tmp_path / "ck"), State(params_ema=params), 5)
    res = str(tmp_path / "res")
    with pytest.raises(RuntimeError):
        run_lib.sample_data(cfg, str(tmp_path / "nope"), res)
    files = run_lib.sample_data(cfg, ckpt, res)
    assert [os.path.basename(f) for f in files] == ["samples_0.npz", "samples_1.npz"]  # 6 // 4 + 1 rounds
    info = np.load(files[0])
    want_keys = {"samples", "nfe_cnt", "samples_x"} | ({"samples_v"} if kind == "cld" else set())
    size = cfg.data.image_size
    assert set(info.files) == want_keys and info["samples"].dtype == np.uint8 and info["samples"].shape == (4, size, size, 3)
    assert int(info["nfe_cnt"]) == 4 and info["samples_x"].shape == (1, 4, size, size, 3)
    assert np.array_equal(info["samples"], np.clip(info["samples_x"][0] * 255.0, 0, 255).astype(np.uint8))
    other = np.load(files[1])
    assert not np.array_equal(other["samples_x"], info["samples_x"])  # a different prior draw per round
    # is_continue skips existing rounds; without it they are regenerated with the same seeds -> identical content
    assert run_lib.sample_data(cfg, ckpt, res, is_continue=True) == []
    again = run_lib.sample_data(cfg, ckpt, res)
    assert np.array_equal(np.load(again[0])["samples_x"], info["samples_x"])
    # sampling_from_fn (cld_jax/run_lib.py:646-671): the caller's own sampling function and state, the same rounds on disk
    if kind == "cld":
        from gddim_amd import datasets
        from gddim_amd.cld import sampling, sde_lib
        from gddim_amd.models import utils as mutils
        sde = sde_lib.from_config(cfg)
        model = mutils.get_model(cfg.model.name)(config=cfg)
        fn = sampling.get_sampling_fn(cfg, sde, model, None, datasets.get_data_inverse_scaler(cfg))
        state = checkpoints.restore_checkpoint(ckpt, State())
        res2 = str(tmp_path / "res2")
        files2 = run_lib.sampling_from_fn(cfg, state, fn, res2)
        assert [os.path.basename(f) for f in files2] == ["samples_0.npz", "samples_1.npz"]
        for a, b in zip(files, files2):
            fa, fb = np.load(a), np.load(b)
            assert set(fa.files) == set(fb.files) and all(np.array_equal(fa[k], fb[k]) for k in fa.files)

