#!/usr/bin/env python
"""Two-dimensional Winograd F(2x2,3x3) fp32 kernel vs the one-dimensional F(2,3) kernel and the direct implicit-GEMM kernel on the
U-Net's 3x3 layer shapes of the 32x32 / 16x16 levels: agreement and time (both Winograd timings include their weight transform).
Usage (GPU box): python tools/bench_wino2d.py [--batch 256] [--iters 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # H, Cin, Cout, count per U-Net eval
    (32, 128, 128, 25), (32, 256, 128, 8), (32, 256, 256, 2), (32, 384, 128, 1),
    (16, 256, 256, 26), (16, 512, 256, 8), (16, 384, 256, 1), (16, 128, 256, 1), (16, 128, 128, 2),
    (8, 256, 256, 29), (8, 512, 256, 9),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only2d", action="store_true", help="time the two-dimensional kernel only (ablation builds: results are wrong by construction)")
    args = ap.parse_args()
    import torch
    from gddim_amd import ops
    dev = torch.device("cuda", 0)
    tot = {"direct": 0.0, "wino1d": 0.0, "wino2d": 0.0}
    tot_fl = 0.0
    for H, Cin, Cout, cnt in SHAPES:
        x = torch.randn(args.batch, H, H, Cin, device=dev)
        w = torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5
        b = torch.randn(Cout, device=dev)
        r = torch.randn(args.batch, H, H, Cout, device=dev)
        u1 = torch.empty(Cout * 12 * Cin, device=dev)
        u2 = torch.empty(Cout * 16 * Cin, device=dev)
        fl = 2.0 * args.batch * H * H * Cout * 9 * Cin
        ref = ops.conv2d(x, w, b, r, out_scale=0.7)
        got = ops.conv3x3_winograd2d(x, w, b, r, out_scale=0.7, u_scratch=u2)
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        line = "H=%2d %3d->%3d  2d vs direct rel max diff %.2e " % (H, Cin, Cout, err)
        variants = (("direct", lambda: ops.conv2d(x, w, b, r, out_scale=0.7)),
                    ("wino1d", lambda: ops.conv3x3_winograd(x, w, b, r, out_scale=0.7, u_scratch=u1)),
                    ("wino2d", lambda: ops.conv3x3_winograd2d(x, w, b, r, out_scale=0.7, u_scratch=u2)))
        for name, fn in (variants[2:] if args.only2d else variants):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            tot[name] += ms * cnt
            line += " %s %7.3f ms %6.1f TF" % (name, ms, fl / ms / 1e9)
        tot_fl += fl * cnt
        print(line, flush=True)
    print("per U-Net eval in these layers: " + ", ".join("%s %.1f ms (%.1f TF alg.)" % (k, v, tot_fl / v / 1e9) for k, v in tot.items() if v > 0))


if __name__ == "__main__":
    main()
