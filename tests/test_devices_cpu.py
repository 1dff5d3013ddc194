"""The device axis of the reference's psampler (cld_jax/sampling.py:232-237, run_lib.py:711-725) on the host side: which devices a
process drives, the per-device keys, the sharded result type, and run_lib's rounds when ONE process drives n_dev engines.  Stand-in
samplers only -- the engines themselves need a GPU (tests/test_gpu_devices.py)."""
import io
import os

import numpy as np
import pytest
import torch

from gddim_amd import devices


def test_local_devices_rules(monkeypatch):
    for k in ("GDDIM_LOCAL_DEVICES", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert devices.local_devices() == [0] and devices.local_device_count() == 1  # no GPU here: one (absent) device, plumbing only
    monkeypatch.setenv("GDDIM_LOCAL_DEVICES", "0,0")  # the same GPU twice: two engines (what the one-GPU test box runs)
    assert devices.local_devices() == [0, 0] and devices.local_device_count() == 2
    monkeypatch.setenv("GDDIM_LOCAL_DEVICES", " 2, 0 ,1 ")
    assert devices.local_devices() == [2, 0, 1]
    for bad in ("a,b", "-1", ","):
        monkeypatch.setenv("GDDIM_LOCAL_DEVICES", bad)
        with pytest.raises(devices.GddimError):
            devices.local_devices()
    monkeypatch.delenv("GDDIM_LOCAL_DEVICES")
    monkeypatch.setenv("WORLD_SIZE", "8")  # a launcher started one process per GPU: this process owns LOCAL_RANK only
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert devices.local_devices() == [5]
    monkeypatch.setenv("GDDIM_LOCAL_DEVICES", "1,3")  # the explicit list wins
    assert devices.local_devices() == [1, 3]


def test_device_seeds():
    from gddim_amd.cld.sampling import _seed_of
    assert devices.device_seeds([7], 1) == [_seed_of(7)] and devices.device_seeds(7, 1) == [7]  # n_dev = 1: the call's own seed
    four = devices.device_seeds([7], 4)
    assert four[0] == 7 and len(set(four)) == 4
    keys = np.array([[0, 11], [0, 12], [0, 13]], dtype=np.uint32)  # a stack of jax-style keys, one per device
    got = devices.device_seeds(keys, 3)
    assert got == [_seed_of(k) for k in keys] and len(set(got)) == 3
    assert devices.device_seeds([5, 6], 2) == [5, 6]


def test_sharded_array_behaves_like_the_pmap_result():
    g = torch.Generator().manual_seed(0)
    shards = [torch.rand((3, 4, 4, 3), generator=g) * 1.2 - 0.1 for _ in range(2)]
    a = devices.ShardedDeviceArray(shards)
    whole = torch.stack(shards).numpy()
    assert a.shape == (2, 3, 4, 4, 3) and a.ndim == 5 and len(a) == 2 and a.dtype == torch.float32
    assert a[1] is shards[1] and torch.equal(a[0, 2], shards[0][2])
    assert np.array_equal(np.asarray(a), whole) and np.array_equal(a.cpu().numpy(), whole) and np.array_equal(a[:, 1], whole[:, 1])
    # the reference's own lines on it, verbatim (cld_jax/run_lib.py:723-730)
    samples_org_x, samples_v, nfe_cnt = a, -a, 50
    samples_x = np.clip(samples_org_x * 255., 0, 255).astype(np.uint8)
    samples_x = samples_x.reshape((-1, 4, 4, 3))
    io_buffer = io.BytesIO()
    np.savez_compressed(io_buffer, samples=samples_x, nfe_cnt=nfe_cnt, samples_v=samples_v, samples_x=samples_org_x)
    io_buffer.seek(0)
    info = np.load(io_buffer)
    assert np.array_equal(info["samples"], np.clip(whole * np.float32(255.0), 0, 255).astype(np.uint8).reshape(-1, 4, 4, 3))
    assert np.array_equal(info["samples_x"], whole) and np.array_equal(info["samples_v"], -whole) and int(info["nfe_cnt"]) == 50
    # arithmetic stays sharded; reductions come back as scalars
    d = (a - a * 0.5).abs()
    assert isinstance(d, devices.ShardedDeviceArray) and abs(float(d.max()) - np.abs(whole * 0.5).max()) < 1e-7
    with pytest.raises(ValueError):
        devices.ShardedDeviceArray([shards[0], shards[1][:2]])


def test_pmap_sampler_axis_plumbing(monkeypatch):
    """psampler's contract with a stand-in sampler: one call per device with that device's slice, key and engine slot; a wrong device
    axis is refused the way pmap refuses it."""
    monkeypatch.setenv("GDDIM_LOCAL_DEVICES", "0,0,0")
    calls = []

    def sampler(rng, state, batch_size, u, slot=0, use_graph=None):
        calls.append((rng, batch_size, slot, use_graph, tuple(u.shape)))
        return u * 2.0, u + 1.0, 7

    u = torch.arange(3 * 2 * 5, dtype=torch.float32).reshape(3, 2, 5)
    xs, vs, nfe = devices.pmap_sampler(sampler, [9, 8, 7], "state", 2, u, None, (5,), 2, use_graph='loop')
    assert [c[0] for c in calls] == [9, 8, 7] and [c[2] for c in calls] == [0, 1, 2] and all(c[1] == 2 and c[3] == 'loop' and c[4] == (2, 5) for c in calls)
    assert nfe == 7 and xs.shape == (3, 2, 5) and np.array_equal(np.asarray(xs), (u * 2.0).numpy()) and np.array_equal(np.asarray(vs), (u + 1.0).numpy())
    with pytest.raises(ValueError, match="device"):
        devices.pmap_sampler(sampler, [0], "state", 2, u[:2], None, (5,), 2)
    # u=None: every device draws its own slice from its own key
    class Sde:
        def prior_sampling(self, seed, shape, device=None):
            return torch.full(shape, float(seed % 1000))
    calls.clear()
    xs, _, _ = devices.pmap_sampler(sampler, [1, 2, 3], "state", 4, None, Sde(), (5,), 2)
    assert xs.shape == (3, 4, 5) and [float(xs[d][0, 0]) for d in range(3)] == [2.0, 4.0, 6.0]


def test_run_lib_rounds_one_process_two_devices(tmp_path, monkeypatch):
    """gddim_amd.run_lib.sample_data with no launcher and two local devices: the sampling function gets eval.batch_size // n_dev and one
    key per device (cld_jax/run_lib.py:715-721), the files hold the reference's layout, writer threads or not."""
    from gddim_amd import distributed, run_lib
    from gddim_amd.cld import sampling
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.models.utils import State
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("GDDIM_LOCAL_DEVICES", "0,0")
    cfg = cld_accr_dcifar10.get_config()
    cfg.data.image_size = 4
    cfg.eval.num_samples, cfg.eval.batch_size = 13, 7  # 2 rounds; 7 // 2 = 3 per device: the remainder is dropped like the reference
    seen = []

    def fake_factory(config, sde, model, shape, inverse_scaler):
        def fn(prng, state, batch_size, u=None):
            seen.append((list(prng), batch_size))
            shards = []
            for k in prng:
                g = torch.Generator().manual_seed(int(k) % (2 ** 31))
                shards.append(torch.rand((batch_size, 4, 4, 3), generator=g))
            xs = devices.ShardedDeviceArray(shards)
            return xs, -xs, 50
        return fn

    monkeypatch.setattr(sampling, "get_sampling_fn", fake_factory)
    blobs = {}
    for writers in (0, 2):
        monkeypatch.setenv("GDDIM_WRITER_THREADS", str(writers))
        seen.clear()
        d = str(tmp_path / ("w%d" % writers))
        stats = {}
        files = run_lib.sample_data(cfg, "unused", d, state=State(params_ema={}), stats=stats)
        assert len(files) == 2 and stats["local_devices"] == 2 and stats["images"] == 2 * 6
        assert seen == [([distributed.rank_seed(43, r, 0), distributed.rank_seed(43, r, 1)], 3) for r in range(2)]
        blobs[writers] = [open(f, "rb").read() for f in files]
        info = np.load(files[1])
        assert info["samples"].shape == (6, 4, 4, 3) and info["samples_x"].shape == (2, 3, 4, 4, 3) and info["samples_v"].shape == (2, 3, 4, 4, 3)
        for dev in range(2):
            g = torch.Generator().manual_seed(distributed.rank_seed(43, 1, dev) % (2 ** 31))
            want = torch.rand((3, 4, 4, 3), generator=g).numpy()
            assert np.array_equal(info["samples_x"][dev], want) and np.array_equal(info["samples_v"][dev], -want)
            assert np.array_equal(info["samples"][3 * dev:3 * dev + 3], np.clip(want * np.float32(255.0), 0, 255).astype(np.uint8))
    assert blobs[0] == blobs[2]


def test_default_graph_is_read_at_call_time():
    """DEFAULT_GRAPH used to be bound as a default argument when the module was imported."""
    import inspect
    from gddim_amd.blur import sampling as bs
    from gddim_amd.cld import sampling as cs
    src = inspect.getsource(cs._make_sampler) + inspect.getsource(bs._make_sampler)
    assert "use_graph=DEFAULT_GRAPH" not in src and src.count("use_graph = DEFAULT_GRAPH") == 2
