#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM kernel on the U-Net's layer shapes (per-shape TFLOP/s via gd_conv2d).
Usage (GPU box): python tools/bench_igemm.py [--batch 256] [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KSWEEP = [(32, c, 128, 3, 0) for c in (32, 64, 128, 256, 384)] + [(16, c, 256, 3, 0) for c in (64, 128, 256, 512)]
SHAPES = [  # H, Cin, Cout, k, count per U-Net eval
    (32, 128, 128, 3, 25), (32, 256, 128, 3, 8), (32, 256, 256, 3, 2), (32, 384, 128, 3, 1),
    (16, 256, 256, 3, 26), (16, 512, 256, 3, 8), (16, 384, 256, 3, 1),
    (8, 256, 256, 3, 29), (8, 512, 256, 3, 9), (4, 256, 256, 3, 31), (4, 512, 256, 3, 9),
    (32, 256, 128, 1, 8), (16, 512, 256, 1, 8), (16, 256, 768, 1, 9), (16, 256, 256, 1, 9), (8, 512, 256, 1, 9),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", type=int, default=-1, help="index into SHAPES")
    ap.add_argument("--ksweep", action="store_true")
    args = ap.parse_args()
    import torch
    from gddim_amd import ops
    dev = torch.device("cuda", 0)
    tot_ms, tot_fl = 0.0, 0.0
    for H, Cin, Cout, k, cnt in (KSWEEP if args.ksweep else (SHAPES if args.only < 0 else [SHAPES[args.only]])):
        x = torch.randn(args.batch, H, H, Cin, device=dev)
        w = torch.randn(Cout, k * k * Cin, device=dev) / (k * k * Cin) ** 0.5
        b = torch.randn(Cout, device=dev)
        for _ in range(2):
            ops.conv2d(x, w, b, None, KH=k, KW=k, stride=1, pad=k // 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            ops.conv2d(x, w, b, None, KH=k, KW=k, stride=1, pad=k // 2)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        fl = 2.0 * args.batch * H * H * Cout * k * k * Cin
        tot_ms += ms * cnt
        tot_fl += fl * cnt
        print("H=%2d %3d->%3d k=%d  M=%6d  %8.3f ms  %6.1f TFLOP/s  (x%d per eval)" %
              (H, Cin, Cout, k, args.batch * H * H, ms, fl / ms / 1e9, cnt), flush=True)
    print("weighted: %.1f ms per U-Net eval in these layers, %.1f TFLOP/s" % (tot_ms, tot_fl / tot_ms / 1e9))


if __name__ == "__main__":
    main()
