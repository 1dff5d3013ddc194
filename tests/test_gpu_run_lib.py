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



def test_sample_data_production_size_pipeline(native_lib, tmp_path, monkeypatch):
    """The host pipeline of the rounds at PRODUCTION size under the driver (cld_jax/run_lib.py:714-731): the full accr
    network, 3 rounds x 256 images, NFE 3.  The pinned-staging / side-stream / writer-thread path must write the serial loop's bytes;
    a writer that fails mid-run surfaces on the caller; `is_continue` then resumes to the same files; and one process driving two
    engines (GDDIM_LOCAL_DEVICES=0,0: 2 x 128 images per round) writes the reference's (n_dev, B / n_dev, H, W, C) layout."""
    import time
    from gddim_amd import run_lib
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.models.utils import State
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GDDIM_LOCAL_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    cfg = cld_accr_dcifar10.get_config()
    cfg.eval.num_samples, cfg.eval.batch_size = 2 * 256 + 1, 256  # 513 // 256 + 1 = 3 rounds
    cfg.sampling.method, cfg.sampling.nfe, cfg.sampling.deis_order, cfg.sampling.noise_removal = 'deis', 3, 2, True
    state = State(params_ema=onet.random_params(cfg, 6, seed=1))
    names = ["samples_%d.npz" % r for r in range(3)]
    blobs, walls = {}, {}
    for writers in (0, 2):
        monkeypatch.setenv("GDDIM_WRITER_THREADS", str(writers))
        d = str(tmp_path / ("w%d" % writers))
        stats = {}
        t0 = time.perf_counter()
        files = run_lib.sample_data(cfg, "unused", d, state=state, stats=stats)
        walls[writers] = time.perf_counter() - t0
        assert [os.path.basename(f) for f in files] == names and sorted(os.listdir(d)) == names
        assert stats["images"] == 3 * 256 and stats["writer_threads"] == writers and stats["local_devices"] == 1
        blobs[writers] = [open(f, "rb").read() for f in files]
    assert blobs[0] == blobs[2], "writer threads changed the archives"
    info = np.load(os.path.join(str(tmp_path / "w2"), "samples_1.npz"))
    assert info["samples"].shape == (256, 32, 32, 3) and info["samples"].dtype == np.uint8 and info["samples_x"].shape == (1, 256, 32, 32, 3)
    assert info["samples_v"].shape == (1, 256, 32, 32, 3) and int(info["nfe_cnt"]) == 3 and np.isfinite(info["samples_x"]).all()
    assert np.array_equal(info["samples"], np.clip(info["samples_x"][0] * 255.0, 0, 255).astype(np.uint8))
    print("production-size rounds: serial %.1f s, 2 writer threads %.1f s (3 x 256 images, NFE 3, engine creation included)" % (walls[0], walls[2]))
    # a writer that raises while round 1 is being written: the caller sees it, round 1 is not on disk, the resume completes the folder
    real = run_lib._RoundWriter._save

    def failing(f_sample, keys):
        if f_sample.endswith("samples_1.npz"):
            raise OSError("disk full (test)")
        real(f_sample, keys)

    d = str(tmp_path / "fail")
    monkeypatch.setattr(run_lib._RoundWriter, "_save", staticmethod(failing))
    with pytest.raises(OSError, match="disk full"):
        run_lib.sample_data(cfg, "unused", d, state=state)
    assert not os.path.exists(os.path.join(d, "samples_1.npz")) and not [f for f in os.listdir(d) if "tmp" in f]
    monkeypatch.setattr(run_lib._RoundWriter, "_save", staticmethod(real))
    again = run_lib.sample_data(cfg, "unused", d, is_continue=True, state=state)
    assert "samples_1.npz" in [os.path.basename(f) for f in again] and sorted(os.listdir(d)) == names
    assert [open(os.path.join(d, n), "rb").read() for n in names] == blobs[0]
    # one process, two engines, no launcher: each engine samples 128 images of a round from its own key (other draws than above, so
    # the check is the layout and the uint8 / float consistency)
    monkeypatch.setenv("GDDIM_LOCAL_DEVICES", "0,0")
    d2 = str(tmp_path / "two")
    stats = {}
    files = run_lib.sample_data(cfg, "unused", d2, state=state, stats=stats)
    assert stats["local_devices"] == 2 and stats["images"] == 3 * 256
    info2 = np.load(files[2])
    assert info2["samples"].shape == (256, 32, 32, 3) and info2["samples_x"].shape == (2, 128, 32, 32, 3) == info2["samples_v"].shape
    assert np.array_equal(info2["samples"], np.clip(info2["samples_x"].reshape(256, 32, 32, 3) * 255.0, 0, 255).astype(np.uint8))
    assert not np.array_equal(info2["samples_x"][0], info2["samples_x"][1])
