#!/usr/bin/env python
"""Headline benchmark: images/sec of the CLD gDDIM sampler (deis_order=2, NFE=50) on CIFAR-10-shaped
synthetic batches -- BASELINE.json `metric`, config[1] at N=1 and config[3] (batch sharded, one process per
GPU, RCCL all-gather of the uint8 images at the end of each round) at N>1.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W
  python bench.py --gpus 2 --dry-run                     (CPU, gloo: the launch / shard / barrier / gather / report path without a GPU)

A "step" is one full sampler call on this rank's batch: 49 x (NCSN++ forward + fused AB update) + the
denoising evaluation + inverse scaling (prior sampling excluded, inputs resident in HBM), then uint8
conversion and, for N>1, the all-gather.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "images/sec at NFE=50 CIFAR-10 32×32 (CLD gDDIM order-2), 1/2/4/8 GPU + FID@50k"
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA
HBM_PEAK_GBS = 8000.0


def hbm_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the newest committed PMC passes (profiles/rNN_hbm_traffic.json:
    (2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 FETCH_SIZE correction), or None when not profiled.  PMC counters need
    their own rocprofv3 passes, so this figure is replayed from the committed profile of the same command, not observed
    in this run."""
    import glob
    table = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            table = json.load(open(path))["kernels"]
            break
        except (OSError, ValueError, KeyError):
            continue
    if table is None:
        return None
    key = kernel_name.replace(" ", "")
    # the engine's profile names abbreviate the template tail of the split-precision kernel (<..., pre> = <..., true, false>)
    keys = {key, key.replace(",pre>", ",true,false,false>"), key.replace(",act>", ",false,false,true>"),
            key.replace("kernel<gn>", "kernel<true>")}
    if key == "conv3x3_halo_bf16x3_kernel":
        keys.add("conv3x3_halo_bf16x3_kernel<false>")
    if key.startswith("igemm_f32_fast_kernel<") and key.count(",") == 4:
        keys.add(key[:-1] + ",false>")  # the default T64 = false template argument rocprofv3 prints
    if key.startswith("igemm_bf16x3_kernel<") and key.count(",") == 3:
        keys.add(key[:-1] + ",false,false,false>")
    for k, v in table.items():
        if k.replace(" ", "") in keys:
            return v["hbm_bytes_per_launch"]
    return None


def issued_ratio(name):
    """MFMA products a kernel issues per algorithmic product of the direct algorithm (by kernel name)."""
    if "wino42" in name:
        return 1.0 / 3.0   # F(2,3)y x F(4,3)x: 24 products per 8 outputs instead of 72
    if "wino2d" in name:
        return 4.0 / 9.0   # F(2x2,3x3): 16 per 4 outputs instead of 36
    if "split6" in name:
        return 24.0 / 9.0  # F(2x2,3x3): 16 components per 2x2 outputs, six bf16 products each, instead of 36
    if "bf16x3" in name:
        return 3.0
    return 1.0


def effective_cpus(cgroup_root="/sys/fs/cgroup"):
    """CPUs this process may really use: the affinity mask and the cgroup CPU quota (v2 cpu.max, v1 cfs_quota / cfs_period) -- a box
    can show 256 logical CPUs to os.cpu_count() and grant far fewer; filling the visible count with threads then only adds throttling."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open(os.path.join(cgroup_root, "cpu.max")).read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open(os.path.join(cgroup_root, "cpu", "cpu.cfs_quota_us")).read())
            per = float(open(os.path.join(cgroup_root, "cpu", "cpu.cfs_period_us")).read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota)))
    return n


def _cpu_worker(idx, threads, cpu_batch, cpu_nfe, image_size, budget_s, barrier, out):
    """One process of the CPU baseline: `cpu_batch` images of their own through the torch-CPU fp32 oracle on `threads` threads.  All
    processes build the same network, warm up, meet at the barrier, then run at the same time.  BOUNDED: the first full-batch network
    evaluation after the barrier is timed and the sampler then runs with as many evaluations as fit `budget_s` (between 2 and `cpu_nfe`),
    so that the leg takes tens of seconds on any box.  (start, end, evaluations) on the shared clock go back."""
    os.environ["OMP_NUM_THREADS"] = str(threads)
    os.environ["MKL_NUM_THREADS"] = str(threads)
    import torch
    torch.set_num_threads(threads)
    from gddim_amd.configs import cld_accr_dcifar10
    from oracle import cld as ocld
    from oracle import ncsnpp as onet
    cfg = cld_accr_dcifar10.get_config()
    cfg.data.image_size = image_size
    params = onet.random_params(cfg, 2 * cfg.data.num_channels, seed=42)  # (timing only: any non-degenerate weights of the architecture)
    osde = ocld.CLD(is_R_rk=cfg.model.is_R_rk, R_dt=cfg.model.R_dt, dtype=np.float64)
    rng = np.random.default_rng(idx)
    H, C = cfg.data.image_size, cfg.data.num_channels
    u = np.stack([rng.standard_normal((cpu_batch, H, H, C)), rng.standard_normal((cpu_batch, H, H, C)) / 2], -1).astype(np.float32)
    eps_fn = onet.cld_eps_fn(params, cfg, dtype=torch.float32)
    eps_fn(u[:1], np.full(1, 0.5, np.float32))  # warm the thread pool / allocator
    barrier.wait(timeout=900)
    t0 = time.time()
    eps_fn(u, np.full(cpu_batch, 0.5, np.float32))  # evaluation 1: also the calibration
    t1 = time.time() - t0
    n = int(max(2, min(cpu_nfe, round((budget_s - t1) / max(t1, 1e-3)))))
    rev_ts = ocld.get_rev_ts(osde, 2, n - 1)
    coef = osde.get_deis_coef(2, rev_ts)
    ocld.deis_sampler(osde, eps_fn, u, n, 2, ts_order=2, denoising=True, inverse_scaler=ocld.inverse_scaler_centered,
                      deis_coef=coef, rev_ts=rev_ts)
    out.put((idx, t0, time.time(), n + 1))


def cpu_baseline(cfg, nfe_full, cpu_batch, cpu_nfe, threads_per_proc=16, budget_s=20.0, procs=0):
    """The reference JAX-CPU path cannot run here (no jax); timed instead: the torch-CPU fp32 ORACLE (a restatement of the same
    algorithm, kind='port') on a BOUNDED sample, on the cores the box really grants (effective_cpus: affinity and cgroup quota --
    the GPU boxes show 256 logical CPUs; filling those with 16 x 16 threads took 414-520 s for this sample in round 6 where ONE
    16-thread process takes 14 s): images are independent, so P processes x T threads each take `cpu_batch` images of their own
    (torch's CPU convolutions collapse when one large pool is thrown at a batch-16 problem, which is why the cores are filled with
    processes, not with one pool).  Every process sizes its own sample to `budget_s` seconds from its first timed evaluation.
    Wall time = last end - first start after a common barrier; value = (images x evaluations done / NFE) / wall."""
    import multiprocessing as mp
    total = os.cpu_count() or 1
    usable = effective_cpus()
    T = max(1, min(threads_per_proc, usable))
    P = max(1, usable // T) if procs <= 0 else procs
    ctx = mp.get_context("spawn")
    barrier, out = ctx.Barrier(P), ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, args=(i, T, cpu_batch, cpu_nfe, cfg.data.image_size, budget_s, barrier, out), daemon=True) for i in range(P)]
    for p_ in procs:
        p_.start()
    import queue
    marks, deadline, why = [], time.time() + 300, None
    while len(marks) < P and why is None:
        try:
            marks.append(out.get(timeout=2))
        except queue.Empty:
            dead = [p_.exitcode for p_ in procs if p_.exitcode not in (None, 0)]
            if dead:
                why = "a worker process exited with code %r" % dead[0]
            elif time.time() > deadline:
                why = "timed out"
    if why is not None:  # a worker died or hung: report what there is instead of losing the bench line (exact processes, started here)
        for p_ in procs:
            if p_.is_alive():
                p_.terminate()
        return {"value": None, "unit": "images/s", "cores": 0, "cores_total": total, "kind": "port",
                "sample": "CPU baseline failed: %d of %d worker processes reported (%s)" % (len(marks), P, why)}
    for p_ in procs:
        p_.join(timeout=60)
    dt = max(m[2] for m in marks) - min(m[1] for m in marks)
    per_proc = sorted(m[2] - m[1] for m in marks)
    evals_pp = sorted(m[3] for m in marks)
    images, evals = P * cpu_batch, cpu_batch * sum(evals_pp)
    return {"value": evals / nfe_full / dt, "unit": "images/s", "cores": P * T, "cores_used": P * T, "cores_total": total, "cores_granted": usable,
            "processes": P, "threads_per_process": T, "kind": "port", "seconds": dt,
            "per_process_seconds": {"min": per_proc[0], "median": per_proc[len(per_proc) // 2], "max": per_proc[-1]},
            "evaluations_per_process": {"min": evals_pp[0], "max": evals_pp[-1]},
            "sample": "torch-CPU fp32 oracle, %d processes x %d threads (%d of the %d logical CPUs are granted to this process), each batch=%d "
                      "through %d..%d network evaluations of the same architecture sized to a %.0f s budget (%d images, %d image-evaluations) "
                      "in %.1f s wall, scaled linearly to NFE=%d"
                      % (P, T, usable, total, cpu_batch, evals_pp[0], evals_pp[-1], budget_s, images, evals, dt, nfe_full)}


def self_launch(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher in the environment: run the same command line as N ranks of one node under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 at a free port) and hand its exit code back.  The reference needs
    no launcher either -- jax.pmap drives the local devices from one process (cld_jax/sampling.py:232-237)."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """--dry-run: everything around the sampler -- rank / world from the environment, the process group (gloo on CPU), the per-rank
    seed, the timed region with its barriers and max-over-ranks, the uint8 conversion, the equal-shard all-gather and rank 0's ONE JSON
    line -- with a stand-in for the sampler call (uniform noise of the image shape).  No GPU, no native library: a plumbing check
    (tests/test_distributed_cpu.py), never a measurement; the line says so."""
    import torch
    from gddim_amd import distributed
    rank, world, _ = distributed.init_from_env(backend="gloo", single_rank_group=True)
    grouped = torch.distributed.is_available() and torch.distributed.is_initialized()
    assert world == args.gpus, "launch with --nproc-per-node equal to --gpus (WORLD_SIZE=%d, --gpus=%d)" % (world, args.gpus)
    B = args.batch
    g = torch.Generator()
    g.manual_seed(distributed.rank_seed(42, 0, rank))

    def step():
        xs = torch.rand((B, args.image_size, args.image_size, 3), generator=g)
        return distributed.all_gather_samples(distributed.to_uint8(xs), equal_shards=True, force=grouped)

    def fence():
        if grouped:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        out = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if grouped:
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
    dt = float(tt.item())
    assert out.shape[0] == B * world and out.dtype == torch.uint8
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * dt / max(1, args.steps), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "dry run: no sampler, no GPU -- launch / shard / gather plumbing only",
                          "dry_run": True, "gathered_images": int(out.shape[0]),
                          "config": {"workload": "dry run of the N-rank harness", "global_batch": B * world,
                                     "parallelism": "dp%d (one process per rank, gloo)" % world}}))
    if grouped:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)   # >= 3 warm-up calls, median of >= 5
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--nfe", type=int, default=50)
    ap.add_argument("--order", type=int, default=2)
    ap.add_argument("--image-size", type=int, default=32)
    ap.add_argument("--workload", choices=["cld", "blur"], default="cld", help="cld = BASELINE configs 2/4/5, blur = config 3")
    ap.add_argument("--precision", choices=["fp32", "bf16x3", "split6"], default="fp32",
                    help="fp32 = fp32 operands on the fp32 matrix core, 3x3 convolutions through the hybrid Winograd transform F(2,3)y x F(4,3)x (headline); bf16x3 = split-precision weight contractions on the bf16 matrix core")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--graph", choices=["step", "loop", "none"], default="loop",
                    help="loop: ONE hipGraph for the whole sampler call (default, the product's default); step: one captured graph per NFE step, replayed; none: eager launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the secondary bf16x3 measurement")
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--cpu-nfe", type=int, default=20, help="network evaluations per image of the CPU sample (batch 16 x 20 evaluations: 10-15 s on 16 threads)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads per process of the CPU baseline; the box is filled with cpu_count // threads processes")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds each CPU-baseline process sizes its sample to")
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the CPU baseline (0: granted CPUs // threads)")
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo plumbing check of the N-rank harness (no sampler, no GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.dry_run:
        return dry_run(args)

    os.environ["GDDIM_PRECISION"] = args.precision  # read at engine creation
    import torch
    from gddim_amd import build, distributed
    from gddim_amd.cld import sampling, sde_lib
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.datasets import get_data_inverse_scaler
    from gddim_amd.models import utils as mutils

    rank, world, local_rank = distributed.init_from_env(single_rank_group=True)
    grouped = torch.distributed.is_available() and torch.distributed.is_initialized()
    assert world == args.gpus, "launch with --nproc-per-node equal to --gpus (WORLD_SIZE=%d, --gpus=%d)" % (world, args.gpus)
    if not os.path.exists(os.path.join(ROOT, "gddim_amd", "libgddim_amd.so")):
        if local_rank == 0:
            build.build()
        if world > 1:
            torch.distributed.barrier()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # this process times ONE GPU (the contract: one process per GPU); without this the sampling function would drive every visible GPU
    # from here, the way the reference's pmap does (gddim_amd/devices.py)
    os.environ["GDDIM_LOCAL_DEVICES"] = str(local_rank)

    B = args.batch
    if args.workload == "cld":
        cfg = cld_accr_dcifar10.get_config()
        cfg.data.image_size = args.image_size
        cfg.sampling.nfe = args.nfe
        cfg.sampling.deis_order = args.order
        cfg.sampling.method = "deis"
        model, _, params = mutils.init_model(42, cfg, init_scale_override=1.0)  # identical weights on every rank
        state = mutils.State(params_ema=params, model_state={})
        sde = sde_lib.from_config(cfg)
        sampling_fn = sampling.get_sampling_fn(cfg, sde, model, None, get_data_inverse_scaler(cfg))  # is_p=True
        shape = (1, B, args.image_size, args.image_size, cfg.data.num_channels)
        g = torch.Generator(device=dev)
        g.manual_seed(distributed.rank_seed(cfg.seed, 0, rank))
        u = torch.stack([torch.randn(shape, generator=g, device=dev), torch.randn(shape, generator=g, device=dev) / 2.0], dim=-1)
        in_ch = 2 * cfg.data.num_channels
        desc = "CLD gDDIM deis_order=%d NFE=%d" % (args.order, args.nfe)
    else:
        from gddim_amd.blur import sampling as bsampling, sde_lib as bsde_lib
        from gddim_amd.configs import blur_ddpm_deep_cifar10
        cfg = blur_ddpm_deep_cifar10.get_config()
        cfg.sampling.nfe = args.nfe
        model, _, params = mutils.init_model(42, cfg, init_scale_override=1.0)
        state = mutils.State(params_ema=params, model_state={})
        sde = bsde_lib.from_config(cfg)
        blur_fn = bsampling.get_sampling_fn(cfg, sde, model, None, get_data_inverse_scaler(cfg))
        g = torch.Generator(device=dev)
        g.manual_seed(distributed.rank_seed(cfg.seed, 0, rank))
        u = torch.randn((1, B, 32, 32, 3), generator=g, device=dev)
        in_ch = cfg.data.num_channels
        desc = "blur gDDIM order0 NFE=%d (ts_order=2, t0=1e-5, sigma_blur_max=10)" % args.nfe

        def sampling_fn(rng, st, bs, uu, use_graph=None):
            xs, n = blur_fn(rng, st, bs, uu, use_graph=use_graph)
            return xs, None, n

    if args.no_graph:
        args.graph = "none"
    graph_mode = {"step": True, "loop": "loop", "none": False}[args.graph]

    def step():
        xs, vs, nfe = sampling_fn([rank], state, B, u, use_graph=graph_mode)
        u8 = distributed.to_uint8(xs[0])
        # equal shards (weak scaling): one all_gather_into_tensor, no size exchange; under a launcher the collective runs
        # through RCCL even at N = 1
        return distributed.all_gather_samples(u8, equal_shards=True, force=grouped)

    def fence():
        if grouped:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    fence()
    # per-step device times from events on the stream the calls are ordered on (recorded, not waited for: the timed region below is
    # exactly K back-to-back steps between two fences, as the contract says); BASELINE.md section 3 quotes the MEDIAN step
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        out = step()
        marks[k + 1].record()
    fence()
    dt = time.perf_counter() - t0
    step_ms = [marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)]
    med_ms = float(np.median(step_ms)) if step_ms else 0.0
    tt = torch.tensor([dt, med_ms], dtype=torch.float64, device=dev)
    if grouped:
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
    dt, med_ms = float(tt[0].item()), float(tt[1].item())
    assert out.shape[0] == B * world and out.dtype == torch.uint8

    eng = model.engine_for(params, in_ch, B, local_rank)
    flops_img = eng.flops_per_image() * args.nfe
    result = {
        "metric": METRIC if (args.workload == "cld" and args.nfe == 50 and args.image_size == 32) else "images/sec, " + desc, "value": world * B / (med_ms * 1e-3) if med_ms > 0 else 0.0, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(1, args.steps), "higher_is_better": True, "scaling": "weak",
        "value_is": "whole-job images per MEDIAN step (BASELINE.md section 3: median of >= 5 after >= 3 warm-up calls; max over ranks of each rank's median)",
        "value_mean": world * B * args.steps / dt, "ms_per_step_median": med_ms, "step_ms": [round(t, 3) for t in step_ms],
        "vs_baseline": None, "dtype": {"fp32": "f32", "bf16x3": "bf16x3 (fp32 split into bf16 hi+lo, fp32 accumulate)", "split6": "f32 (3x3 convolutions: exact three-piece bf16 split, fp32 accumulate)"}[args.precision], "data": "synthetic",
        "config": {"workload": "%s, batch=%d per GPU, %dx%dx3 synthetic noise, random-init NCSN++ "
                               "(nf=128, ch_mult=(1,2,2,2), 8 res blocks, attn@16)" % (desc, B, cfg.data.image_size, cfg.data.image_size),
                   "global_batch": B * world, "parallelism": "dp%d (one process per GPU, uint8 all-gather per round%s)" % (world, ", RCCL" if grouped else ""),
                   "hipgraph": {True: "per-step graph replayed NFE-1 times + denoise graph", "loop": "one graph per sampler call", False: False}[graph_mode],
                   "tflop_per_image": flops_img / 1e12, "tflop_per_image_is": "the reference's algorithmic count (direct convolutions, four attention projections)"},
        "e2e_tflops_per_gpu": B * args.steps * flops_img / dt / 1e12,
    }

    # ---- roofline of the dominant kernel: live hipEvent timing of every launch of one more sampler call ----
    if not args.no_profile:
        eng.profile(True)
        sampling_fn([rank], state, B, u, use_graph=False)
        torch.cuda.synchronize()
        prof = eng.profile_read()
        eng.profile(False)
        tot_ms = sum(p["ms"] for p in prof)
        dom = max(prof, key=lambda p: p["ms"])
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        # bf16x3 issues 3 bf16 MFMAs per algorithmic product: achieved stays ALGORITHMIC flops, the peak is the bf16 one
        peak = BF16_MFMA_PEAK_TFLOPS if "bf16x3" in dom["name"] else FP32_MFMA_PEAK_TFLOPS
        # `achieved` = ALGORITHMIC flops of the direct algorithm / time.  `frac` = fraction of the matrix
        # pipe the kernel keeps busy = ISSUED MFMA flops / time / peak (<= 1): the hybrid Winograd kernel issues 1/3 of the algorithmic
        # products, bf16x3 issues 3 bf16 MFMAs per algorithmic product.  `algorithmic_frac` = achieved / peak (may exceed 1
        # for Winograd: the kernel does less work than the direct algorithm).
        issue_ratio = issued_ratio(dom["name"])
        traffic = hbm_traffic(dom["name"])
        alg_bytes = dom["bytes"] / max(1, dom["launches"]) if dom.get("bytes") else None
        result["roofline"] = {"kernel": dom["name"], "bound": "mfma", "achieved": achieved, "peak": peak,
                              "unit": "TFLOP/s", "frac": achieved * issue_ratio / peak, "traffic": traffic,
                              "algorithmic_bytes": alg_bytes, "traffic_ratio": (traffic / alg_bytes) if (traffic and alg_bytes) else None,
                              "algorithmic_frac": achieved / peak, "issued": achieved * issue_ratio,
                              "launches": dom["launches"], "avg_launch_ms": dom["ms"] / max(1, dom["launches"]),
                              "flops_per_launch": dom["flops"] / max(1, dom["launches"]),
                              "share_of_gpu_time": dom["ms"] / tot_ms if tot_ms else None,
                              "traffic_source": "committed rocprofv3 --pmc passes of this command (profiles/), not this run",
                              "timing_source": "one extra EAGER sampler call under the engine's hipEvent profiler (events around every launch on "
                                               "the launch stream) -- not the hipGraph replay that `value` times; rocprofv3's average for the "
                                               "same kernel in the replayed run is in profiles/"}
        # what the matrix pipe really multiplies per step (Winograd kernels issue 1/3 or 4/9 of the direct algorithm's products, the folded
        # attention projections and the up-sampling shortcuts' high-resolution share are not multiplied at all): next to the headline's
        # reference-algorithmic TFLOP/s (`e2e_tflops_per_gpu` prices the reference's 37.2 GFLOP per image and evaluation)
        issued = sum(p["flops"] * issued_ratio(p["name"]) for p in prof if p["flops"] and "bf16x3" not in p["name"])
        result["e2e_issued_fp32_tflops_per_gpu"] = issued / (dt / max(1, args.steps)) / 1e12
        result["e2e_issued_frac_of_fp32_mfma_peak"] = result["e2e_issued_fp32_tflops_per_gpu"] / FP32_MFMA_PEAK_TFLOPS
        result["kernel_breakdown_source"] = "the same eager profiling pass (its sum exceeds ms_per_step by the launch gaps the graph replay hides)"
        if "wino42" in dom["name"]:
            result["roofline"]["note"] = ("hybrid Winograd F(2,3)y x F(4,3)x: achieved = algorithmic flops of the direct convolution / time; the kernel "
                                          "issues 24 products per 2x4 output tile instead of 72, i.e. 1/3 of that; frac = issued / dense fp32 MFMA peak "
                                          "(nominal, 2.4 GHz).  fp32 error of the transform: profiles/r03_winograd_error_study.md")
        elif "wino2d" in dom["name"]:
            result["roofline"]["note"] = ("Winograd F(2x2,3x3): achieved = algorithmic flops of the direct convolution / time; the kernel "
                                          "issues 16 products per 2x2 output quad instead of 36, i.e. 4/9 of that; frac = issued / dense fp32 MFMA peak (nominal, 2.4 GHz); "
                                          "the in-kernel timeline (profiles/r02_notes.md, tools/wino2d_trace.py) shows the shader clock at 1.9-2.1 GHz under this "
                                          "kernel (power-limited) and its K loop at 82-88 % of the MFMA issue bound in shader cycles")
        result["kernel_breakdown"] = [
            {"name": p["name"], "ms": round(p["ms"], 3), "launches": p["launches"],
             "tflops": (p["flops"] / (p["ms"] * 1e-3) / 1e12) if p["flops"] and p["ms"] else None,
             # fraction of the dense fp32 MFMA peak in ISSUED products (Winograd kernels issue 1/3, 4/9, 2/3 of the algorithmic flops)
             "issued_frac": (p["flops"] * issued_ratio(p["name"]) / (p["ms"] * 1e-3) / 1e12 / (BF16_MFMA_PEAK_TFLOPS if "bf16x3" in p["name"] else FP32_MFMA_PEAK_TFLOPS)) if p["flops"] and p["ms"] else None,
             "gbs": (p["bytes"] / (p["ms"] * 1e-3) / 1e9) if p["bytes"] and p["ms"] else None}
            for p in sorted(prof, key=lambda p: -p["ms"])]

    def timed(nsteps, mode):
        """nsteps synchronised sampler calls in hipGraph mode `mode` after one untimed call; max over ranks (secondary measurements)."""
        def one():
            xs, vs, nfe = sampling_fn([rank], state, B, u, use_graph=mode)
            return distributed.all_gather_samples(distributed.to_uint8(xs[0]), equal_shards=True, force=grouped)
        one()
        fence()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            one()
        fence()
        tx = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if grouped:
            torch.distributed.all_reduce(tx, op=torch.distributed.ReduceOp.MAX)
        return float(tx.item())

    # ---- the two captured forms of the loop next to each other (north-star: "the outer NFE loop captured as a hipGraph"): per-step graph
    # replayed NFE - 1 times vs ONE graph for the whole sampler call; bit-identical outputs (tests/test_gpu_sampler.py), bounded sample ----
    side_steps = max(1, min(args.steps, 3))
    if not args.no_alt and graph_mode is not False:
        ms = {}
        for name, mode in (("step", True), ("loop", "loop")):
            ms[name] = 1e3 * timed(side_steps, mode) / side_steps
        result["graph_modes"] = {"ms_per_step": ms, "steps": side_steps, "headline_uses": args.graph,
                                 "note": "per-step graph replayed NFE-1 times vs one graph per sampler call, same kernels, bit-identical outputs"}

    # ---- secondary measurement: the opt-in split-precision mode on the same workload (never the headline `value`; bounded sample) ----
    if args.precision == "fp32" and not args.no_alt:
        eng.set_precision("bf16x3")
        dta = timed(side_steps, graph_mode)
        alt = {"dtype": "bf16x3 (fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate)",
               "value": world * B * side_steps / dta, "unit": "images/s", "ms_per_step": 1e3 * dta / side_steps, "steps": side_steps,
               "e2e_algorithmic_tflops_per_gpu": B * side_steps * flops_img / dta / 1e12,
               "parity": "same tests and bars as fp32 (tests/test_gpu_bf16x3.py)"}
        if not args.no_profile:
            eng.profile(True)
            sampling_fn([rank], state, B, u, use_graph=False)
            torch.cuda.synchronize()
            prof = eng.profile_read()
            eng.profile(False)
            dom = max(prof, key=lambda p: p["ms"])
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
            alt["roofline"] = {"kernel": dom["name"], "bound": "mfma", "achieved": ach, "peak": BF16_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s (algorithmic; the kernel issues 3x this in bf16 MFMAs)",
                               "frac": 3 * ach / BF16_MFMA_PEAK_TFLOPS, "algorithmic_frac": ach / BF16_MFMA_PEAK_TFLOPS, "issued": 3 * ach,
                               "traffic": hbm_traffic(dom["name"]), "launches": dom["launches"],
                               "avg_launch_ms": dom["ms"] / max(1, dom["launches"])}
        eng.set_precision("fp32")
        result["bf16x3"] = alt
        # ---- second secondary measurement (round 6): split6 mode -- fp32 mode with the 3x3 convolutions of the 32x32 / 16x16 levels through
        # F(2x2,3x3) in the exact three-piece bf16 split on the bf16 matrix core (fp32-class error: tests/test_gpu_split6*.py); never the
        # headline `value` this round ----
        eng.set_precision("split6")
        dts = timed(side_steps, graph_mode)
        alt6 = {"dtype": "f32 operands; 3x3 convolutions of the 16x16-tileable levels as F(2x2,3x3) in the exact three-piece bf16 split (six bf16 MFMAs per "
                         "product, fp32 accumulate), everything else fp32 MFMA",
                "value": world * B * side_steps / dts, "unit": "images/s", "ms_per_step": 1e3 * dts / side_steps, "steps": side_steps,
                "e2e_algorithmic_tflops_per_gpu": B * side_steps * flops_img / dts / 1e12,
                "parity": "same oracles and bars as fp32 (tests/test_gpu_split6_mode.py); per-layer rms error 0.33 of the fp32 hybrid Winograd kernel's (tests/test_gpu_split6.py)"}
        if not args.no_profile:
            eng.profile(True)
            sampling_fn([rank], state, B, u, use_graph=False)
            torch.cuda.synchronize()
            prof = eng.profile_read()
            eng.profile(False)
            k6 = [p_ for p_ in prof if "split6" in p_["name"]]
            if k6:
                k6 = {"name": "conv3x3_wino2d_split6_kernel (all launches, with and without the epilogue GroupNorm)", "ms": sum(q["ms"] for q in k6),
                      "flops": sum(q["flops"] for q in k6), "launches": sum(q["launches"] for q in k6)}
                ach = k6["flops"] / (k6["ms"] * 1e-3) / 1e12 if k6["ms"] > 0 else 0.0
                alt6["roofline"] = {"kernel": k6["name"], "bound": "mfma", "achieved": ach, "peak": BF16_MFMA_PEAK_TFLOPS,
                                    "unit": "TFLOP/s (algorithmic; the kernel issues 24/9 of this in bf16 MFMAs)",
                                    "frac": ach * 24.0 / 9.0 / BF16_MFMA_PEAK_TFLOPS, "algorithmic_frac": ach / BF16_MFMA_PEAK_TFLOPS,
                                    "launches": k6["launches"], "avg_launch_ms": k6["ms"] / max(1, k6["launches"]),
                                    "share_of_gpu_time": k6["ms"] / max(1e-9, sum(p_["ms"] for p_ in prof))}
            alt6["kernel_breakdown"] = [{"name": p_["name"], "ms": round(p_["ms"], 3), "launches": p_["launches"]} for p_ in sorted(prof, key=lambda q: -q["ms"])[:8]]
        eng.set_precision("fp32")
        result["split6"] = alt6

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "cld":
        result["cpu_baseline"] = cpu_baseline(cfg, args.nfe, args.cpu_batch, args.cpu_nfe, args.cpu_threads, args.cpu_budget, args.cpu_procs)
    if rank == 0:
        print(json.dumps(result))
    if grouped:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
