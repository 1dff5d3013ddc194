"""N > 1 host logic on CPU: world_size-2 gloo processes exercise batch sharding, per-rank seeds and the end-of-round
all-gather of uint8 samples (the only exchange on the path)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, global_batch, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from gddim_amd import distributed
    r, w, lr = distributed.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    start, size = distributed.shard_batch(global_batch, w, r)
    # stand-in for this rank's sampler output: image i of the round holds the value i/255 everywhere
    x = (torch.arange(start, start + size, dtype=torch.float32) / 255.0).reshape(-1, 1, 1, 1).expand(size, 4, 4, 3)
    u8 = distributed.to_uint8(x + 0.9 / 255.0)  # truncation, not rounding
    full = distributed.all_gather_samples(u8.contiguous())
    np.save(os.path.join(out_dir, "rank%d.npy" % r), full.numpy())
    # equal shards: one collective, no size exchange; and the same answer as the general path
    eq = torch.full((3, 2, 2, 3), r, dtype=torch.uint8)
    a = distributed.all_gather_samples(eq, equal_shards=True)
    b = distributed.all_gather_samples(eq)
    assert torch.equal(a, b) and a.shape[0] == 3 * w and all(int(a[3 * k, 0, 0, 0]) == k for k in range(w))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_shard_and_all_gather_gloo(tmp_path):
    world, global_batch = 2, 7  # uneven: shards of 4 and 3
    mp.spawn(_worker, args=(world, _free_port(), global_batch, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npy" % r)) for r in range(world)]
    assert outs[0].shape == (global_batch, 4, 4, 3) and outs[0].dtype == np.uint8
    assert np.array_equal(outs[0], outs[1])                                   # every rank holds the whole round
    assert np.array_equal(outs[0][:, 0, 0, 0], np.arange(global_batch))       # order = rank order, values truncated


def test_shard_batch_and_seeds():
    from gddim_amd import distributed
    for gb, w in ((2048, 8), (1024, 8), (7, 2), (5, 8)):
        parts = [distributed.shard_batch(gb, w, r) for r in range(w)]
        assert sum(s for _, s in parts) == gb
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        assert max(s for _, s in parts) - min(s for _, s in parts) <= 1
    seeds = {distributed.rank_seed(42, rnd, r) for rnd in range(50) for r in range(8)}
    assert len(seeds) == 400
    assert np.array_equal(distributed.to_uint8(np.array([-0.1, 0.0, 0.999, 0.5, 1.2])), np.array([0, 0, 254, 127, 255], np.uint8))


def _round_worker(rank, world, port, out_dir, batch=6):
    """run_lib.sample_data under two gloo ranks with the sampler factory replaced by a CPU stand-in: what is exercised is the
    round loop itself -- per-rank batch, per-(round, rank) seeds, the three all-gathers, rank 0 writing the files."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from gddim_amd import distributed, run_lib
    from gddim_amd.cld import sampling
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.models.utils import State
    distributed.init_from_env(backend="gloo")
    cfg = cld_accr_dcifar10.get_config()
    cfg.data.image_size = 4
    cfg.eval.num_samples, cfg.eval.batch_size = 10, batch  # 2 rounds of 6 images, 3 per rank (7: 4 + 3)
    calls = []

    def fake_factory(config, sde, model, shape, inverse_scaler):
        def fn(rng, state, batch_size, u=None):
            calls.append((int(rng[0]), batch_size))
            g = torch.Generator().manual_seed(int(rng[0]) % (2 ** 31))
            x = torch.rand((1, batch_size, 4, 4, 3), generator=g)
            return x, -x, 50
        return fn

    sampling.get_sampling_fn = fake_factory
    written = run_lib.sample_data(cfg, "unused", out_dir, state=State(params_ema={}))
    seeds = [distributed.rank_seed(cfg.seed + 1, r, rank) for r in range(2)]
    per = distributed.shard_batch(batch, world, rank)[1]
    assert calls == [(seeds[0], per), (seeds[1], per)]
    assert (len(written) == 2) == (rank == 0)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_sample_data_rounds_two_ranks_gloo(tmp_path):
    from gddim_amd import distributed
    mp.spawn(_round_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        info = np.load(os.path.join(str(tmp_path), "samples_%d.npz" % r))
        assert info["samples"].shape == (6, 4, 4, 3) and info["samples"].dtype == np.uint8 and int(info["nfe_cnt"]) == 50
        # rank order inside the round, each rank's three images drawn from its own (round, rank) seed
        want = []
        for rank in range(2):
            g = torch.Generator().manual_seed(distributed.rank_seed(42 + 1, r, rank) % (2 ** 31))
            want.append(torch.rand((1, 3, 4, 4, 3), generator=g)[0])
        want = torch.cat(want).numpy()
        # float copies in the reference's (n_dev, B / n_dev, H, W, C) layout (cld_jax/run_lib.py:722-730)
        assert info["samples_x"].shape == (2, 3, 4, 4, 3) and info["samples_v"].shape == (2, 3, 4, 4, 3)
        assert np.array_equal(info["samples_x"].reshape(want.shape), want) and np.array_equal(info["samples_v"].reshape(want.shape), -want)
        assert np.array_equal(info["samples"], np.clip(want * 255.0, 0, 255).astype(np.uint8))


def test_sample_data_uneven_round_keeps_every_image(tmp_path):
    """eval.batch_size = 7 over two ranks: shards of 4 and 3; the round file still holds 7 images (the reference would drop one)."""
    mp.spawn(_round_worker, args=(2, _free_port(), str(tmp_path), 7), nprocs=2, join=True)
    info = np.load(os.path.join(str(tmp_path), "samples_0.npz"))
    assert info["samples"].shape == (7, 4, 4, 3) and info["samples_x"].shape == (1, 7, 4, 4, 3)


def test_bench_self_launches_for_n_gt_1(tmp_path):
    """`python bench.py --gpus 2` with NO launcher in the environment (how the driver calls it for N = 1) must start its own two ranks
    (it used to die on `assert world == args.gpus`).  --dry-run keeps the GPU and the sampler out of it: gloo ranks, the
    timed region with its barriers, the uint8 all-gather and rank 0's single JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1",
                        "--batch", "8"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines  # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None and d["steps"] == 2 and d["warmup"] == 1
    assert d["gathered_images"] == 16 and d["config"]["global_batch"] == 16 and d["scaling"] == "weak"
    # under an external launcher the same file still works (the driver's N > 1 form): torch.distributed.run sets RANK, no re-exec
    port = _free_port()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0",
                        "--batch", "4"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    d = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["gathered_images"] == 8


def _fake_rounds_setup(monkeypatch_target, n_rounds_samples=25, batch=6, size=4):
    from gddim_amd.cld import sampling
    from gddim_amd.configs import cld_accr_dcifar10
    cfg = cld_accr_dcifar10.get_config()
    cfg.data.image_size = size
    cfg.eval.num_samples, cfg.eval.batch_size = n_rounds_samples, batch

    def fake_factory(config, sde, model, shape, inverse_scaler):
        def fn(rng, state, batch_size, u=None):
            g = torch.Generator().manual_seed(int(rng[0]) % (2 ** 31))
            x = torch.rand((1, batch_size, size, size, 3), generator=g)
            return x, -x, 50
        return fn

    monkeypatch_target.setattr(sampling, "get_sampling_fn", fake_factory)
    return cfg


def test_round_writer_threads_write_the_serial_files(tmp_path, monkeypatch):
    """run_lib's rounds with the host work (copies, zlib, file) on writer threads against the serial loop the
    reference has (cld_jax/run_lib.py:714-731): same files, same keys, byte-identical archives; resume (`is_continue`) skips what
    exists and rewrites what is missing; no `.tmp` file is left behind; a failing write surfaces on the caller's thread."""
    from gddim_amd import run_lib
    from gddim_amd.models.utils import State
    cfg = _fake_rounds_setup(monkeypatch)
    st = State(params_ema={})
    outs = {}
    for writers in (0, 1, 3):
        monkeypatch.setenv("GDDIM_WRITER_THREADS", str(writers))
        d = str(tmp_path / ("w%d" % writers))
        stats = {}
        files = run_lib.sample_data(cfg, "unused", d, state=st, stats=stats)
        assert [os.path.basename(f) for f in files] == ["samples_%d.npz" % r for r in range(25 // 6 + 1)]
        assert sorted(os.listdir(d)) == sorted(os.path.basename(f) for f in files)  # no temporary left
        assert stats["images"] == 5 * 6 and stats["writer_threads"] == writers and len(stats["host_submit_s"]) == 5
        outs[writers] = [open(f, "rb").read() for f in files]
    assert outs[0] == outs[1] == outs[3]
    info = np.load(os.path.join(str(tmp_path / "w3"), "samples_2.npz"))
    assert set(info.files) == {"samples", "nfe_cnt", "samples_v", "samples_x"} and info["samples"].shape == (6, 4, 4, 3)
    assert np.array_equal(info["samples"], np.clip(info["samples_x"][0] * 255.0, 0, 255).astype(np.uint8))
    # resume: remove two rounds, only those come back, identical
    d = str(tmp_path / "w3")
    os.remove(os.path.join(d, "samples_1.npz"))
    os.remove(os.path.join(d, "samples_4.npz"))
    again = run_lib.sample_data(cfg, "unused", d, is_continue=True, state=st)
    assert [os.path.basename(f) for f in again] == ["samples_1.npz", "samples_4.npz"]
    assert [open(os.path.join(d, "samples_%d.npz" % r), "rb").read() for r in range(5)] == outs[0]
    # the config's own knob wins over the environment
    cfg.eval.writer_threads = 0
    stats = {}
    run_lib.sample_data(cfg, "unused", str(tmp_path / "cfgknob"), state=st, stats=stats)
    assert stats["writer_threads"] == 0


def test_round_writer_error_reaches_the_caller(tmp_path, monkeypatch):
    import pytest
    from gddim_amd import run_lib
    from gddim_amd.models.utils import State
    cfg = _fake_rounds_setup(monkeypatch)
    monkeypatch.setenv("GDDIM_WRITER_THREADS", "2")
    real = run_lib._RoundWriter._save

    def failing(f_sample, keys):
        if f_sample.endswith("samples_1.npz"):
            raise OSError("disk full (test)")
        real(f_sample, keys)

    monkeypatch.setattr(run_lib._RoundWriter, "_save", staticmethod(failing))
    with pytest.raises(OSError, match="disk full"):
        run_lib.sample_data(cfg, "unused", str(tmp_path / "e"), state=State(params_ema={}))
    assert not os.path.exists(os.path.join(str(tmp_path / "e"), "samples_1.npz"))


def test_bench_cpu_baseline_counts_the_granted_cpus(tmp_path):
    """bench.effective_cpus: the CPU baseline fills the CPUs the box GRANTS, not the ones it shows (round 6: the GPU boxes show 256 logical
    CPUs under a cgroup quota of 16; 16 x 16 threads took 414-520 s for a 14 s sample) -- cgroup v2 `cpu.max`, v1 quota / period, no quota."""
    sys.path.insert(0, ROOT)
    import bench
    n = os.cpu_count()
    v2 = tmp_path / "v2"
    v2.mkdir()
    (v2 / "cpu.max").write_text("300000 100000\n")
    assert bench.effective_cpus(str(v2)) == min(3, n)
    (v2 / "cpu.max").write_text("max 100000\n")
    assert bench.effective_cpus(str(v2)) == min(n, len(os.sched_getaffinity(0)))
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("250000\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert bench.effective_cpus(str(v1)) == min(2, n)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert bench.effective_cpus(str(v1)) == min(n, len(os.sched_getaffinity(0)))
    assert 1 <= bench.effective_cpus(str(tmp_path / "absent")) <= n
