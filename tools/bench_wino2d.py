#!/usr/bin/env python
"""The fp32 3x3 convolution kernels on the U-Net's 3x3 layer shapes of the 32x32 / 16x16 / 8x8 levels: hybrid Winograd F(2,3)y x F(4,3)x
(round 3), two-dimensional F(2x2,3x3) (round 2) and the direct implicit-GEMM kernel: agreement with the
direct kernel, error against float64 on the first image, and time (the Winograd timings include their weight transform).
Usage (GPU box): python tools/bench_wino2d.py [--batch 256] [--iters 10] [--variants wino42,wino2d]"""
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
    ap.add_argument("--variants", default="direct,wino2d,wino42")
    args = ap.parse_args()
    import torch
    from gddim_amd import ops
    dev = torch.device("cuda", 0)
    tot = {"direct": 0.0, "wino2d": 0.0, "wino42": 0.0}
    want_v = ["wino2d"] if args.only2d else args.variants.split(",")
    import torch.nn.functional as F
    tot_fl = 0.0
    for H, Cin, Cout, cnt in SHAPES:
        x = torch.randn(args.batch, H, H, Cin, device=dev)
        w = torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5
        b = torch.randn(Cout, device=dev)
        r = torch.randn(args.batch, H, H, Cout, device=dev)
        u2 = torch.empty(Cout * 16 * Cin, device=dev)
        u3 = torch.empty(24 * ((Cout + 31) // 32 * 32) * Cin, device=dev)
        fl = 2.0 * args.batch * H * H * Cout * 9 * Cin
        ref = ops.conv2d(x, w, b, r, out_scale=0.7)
        # float64 reference on the first image (CPU)
        w64 = w.cpu().double().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        want = ((F.conv2d(x[:1].cpu().double().permute(0, 3, 1, 2), w64, padding=1).permute(0, 2, 3, 1) + b.cpu().double() + r[:1].cpu().double()) * 0.7)
        variants = (("direct", lambda: ops.conv2d(x, w, b, r, out_scale=0.7)),
                    ("wino2d", lambda: ops.conv3x3_winograd2d(x, w, b, r, out_scale=0.7, u_scratch=u2)),
                    ("wino42", lambda: ops.conv3x3_winograd42(x, w, b, r, out_scale=0.7, u_scratch=u3)))
        variants = [v for v in variants if v[0] in want_v]
        line = "H=%2d %3d->%3d " % (H, Cin, Cout)
        for name, fn in variants:
            got = fn()
            line += " %s: vs direct %.1e, vs f64 %.1e |" % (name, float((got - ref).abs().max()) / float(ref.abs().max()),
                                                          float((got[:1].cpu().double() - want).abs().max()) / float(want.abs().max()))
        for name, fn in variants:
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
