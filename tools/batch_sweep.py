#!/usr/bin/env python
"""Images/s of the CLD gDDIM sampler (deis_order 2, NFE 50, fp32, per-step hipGraph) as a function of the per-GPU batch, with the
kernels every level of the U-Net ran on at that batch -- the reference's own per-device batch is eval.batch_size / local_device_count
= 1024 / 8 = 128 (cld_jax/run_lib.py:721, configs/default_cifar10_config.py:54), BASELINE config 2 uses 256.

  python tools/batch_sweep.py [--batches 16,32,64,128,192,256,320] [--nfe 50] [--out profiles/r04_batch_sweep.json]

Per batch: one warm-up call, `--calls` timed calls (wall clock around synchronised calls), then ONE eager U-Net evaluation under the
engine's profiler with GD_PROF_BY_LABEL=2 keys ("label class|kernel"), folded into  level (H) -> kernel -> ms, launches.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GD_PROF_BY_LABEL", "2")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="16,32,64,128,192,256,320")
    ap.add_argument("--nfe", type=int, default=50)
    ap.add_argument("--calls", type=int, default=2)
    ap.add_argument("--image-size", type=int, default=32)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "batch_sweep.json"))
    args = ap.parse_args()
    import torch
    from gddim_amd.cld import sampling, sde_lib
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.datasets import get_data_inverse_scaler
    from gddim_amd.models import utils as mutils
    dev = torch.device("cuda", 0)
    cfg = cld_accr_dcifar10.get_config()
    cfg.data.image_size = args.image_size
    cfg.sampling.nfe, cfg.sampling.deis_order, cfg.sampling.method = args.nfe, 2, "deis"
    model, _, params = mutils.init_model(42, cfg, init_scale_override=1.0)
    state = mutils.State(params_ema=params, model_state={})
    sde = sde_lib.from_config(cfg)
    fn = sampling.get_sampling_fn(cfg, sde, model, None, get_data_inverse_scaler(cfg))
    rows = []
    for B in [int(b) for b in args.batches.split(",")]:
        g = torch.Generator(device=dev)
        g.manual_seed(B)
        shape = (1, B, args.image_size, args.image_size, 3)
        u = torch.stack([torch.randn(shape, generator=g, device=dev), torch.randn(shape, generator=g, device=dev) / 2.0], dim=-1)
        fn([0], state, B, u)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.calls):
            fn([0], state, B, u)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.calls
        eng = model.engine_for(params, 6, B, 0)
        eng.profile(True)
        xin = torch.randn((B, args.image_size, args.image_size, 6), device=dev)
        eng.unet(xin, torch.full((B,), 500.0, device=dev))
        torch.cuda.synchronize()
        prof = eng.profile_read()
        eng.profile(False)
        levels = {}
        for p in prof:
            name = p["name"]
            m = re.search(r"@(\d+)x(\d+):", name)
            kern = name.split("|", 1)[1] if "|" in name else name
            is3 = ".conv" in name and "conv2" not in name  # Conv_0 / Conv_1 of the residual blocks + stem / head / pyramid
            key = ("%sx%s" % (m.group(1), m.group(2))) if m else "-"
            d = levels.setdefault(key, {}).setdefault(kern, {"ms": 0.0, "launches": 0, "gflop": 0.0, "conv3x3_ms": 0.0})
            d["ms"] += p["ms"]
            d["launches"] += p["launches"]
            d["gflop"] += p["flops"] / 1e9
            if is3:
                d["conv3x3_ms"] += p["ms"]
        for lv in levels.values():
            for d in lv.values():
                d["ms"], d["gflop"], d["conv3x3_ms"] = round(d["ms"], 4), round(d["gflop"], 2), round(d["conv3x3_ms"], 4)
        row = {"batch": B, "images_per_s": B / dt, "s_per_call": dt, "ms_per_image": 1e3 * dt / B,
               "unet_eval_ms_eager_profiled": round(sum(p["ms"] for p in prof), 3), "levels": levels}
        rows.append(row)
        print("B=%4d  %8.2f images/s  %.3f s per call  (one eager evaluation: %.2f ms of kernels)" % (B, row["images_per_s"], dt, row["unet_eval_ms_eager_profiled"]), flush=True)
        for key in sorted(levels, key=lambda k: -int(k.split("x")[0]) if k != "-" else 1):
            for kern, d in sorted(levels[key].items(), key=lambda kv: -kv[1]["ms"])[:4]:
                print("        %-6s %-60s %8.3f ms %4d launches %s" % (key, kern[:60], d["ms"], d["launches"],
                                                                        ("%6.1f TFLOP/s" % (d["gflop"] / d["ms"])) if d["gflop"] and d["ms"] else ""), flush=True)
    best = max(r["images_per_s"] for r in rows)
    out = {"workload": "CLD gDDIM deis_order=2 NFE=%d, %dx%dx3, fp32, per-step hipGraph, full accr NCSN++" % (args.nfe, args.image_size, args.image_size),
           "note": "images_per_s: wall clock over %d synchronised sampler calls after one warm-up; levels: one EAGER U-Net evaluation under the "
                   "engine's hipEvent profiler (label class|kernel keys)" % args.calls,
           "rows": rows, "relative_to_best": {str(r["batch"]): r["images_per_s"] / best for r in rows}}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
