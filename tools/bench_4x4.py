#!/usr/bin/env python
"""The 4x4 level's 3x3 convolutions in isolation (B x 4 x 4, 256 -> 256 and 512 -> 256): F(2x2,3x3) with the channel range split over
workgroups + finishing pass (GDDIM_WINO2_KS forces the factor), against the direct implicit-GEMM kernel.  Run under
`rocprofv3 --kernel-trace --stats` to see the main kernel and the finishing pass apart.   python tools/bench_4x4.py [--batch 256]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    import torch
    from gddim_amd import ops
    dev = torch.device("cuda", 0)
    for Cin, Cout in ((256, 256), (512, 256)):
        x = torch.randn(args.batch, 4, 4, Cin, device=dev)
        w = torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5
        b = torch.randn(Cout, device=dev)
        r = torch.randn(args.batch, 4, 4, Cout, device=dev)
        u2 = torch.empty(Cout * 16 * Cin, device=dev)
        ref = ops.conv2d(x, w, b, r, out_scale=0.7)
        fl = 2.0 * args.batch * 16 * Cout * 9 * Cin
        for name, fn in (("direct", lambda: ops.conv2d(x, w, b, r, out_scale=0.7)),
                         ("wino2d", lambda: ops.conv3x3_winograd2d(x, w, b, r, out_scale=0.7, u_scratch=u2))):
            got = fn()
            err = float((got - ref).abs().max()) / float(ref.abs().max())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.iters
            print("4x4 B=%d %d->%d %-7s %7.1f us  %6.1f TFLOP/s alg.  (vs direct %.1e)%s" % (
                args.batch, Cin, Cout, name, us, fl / us / 1e6, err, "  [includes the weight transform]" if name == "wino2d" else ""), flush=True)


if __name__ == "__main__":
    main()
