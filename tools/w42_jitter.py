"""Per-launch timing of the hybrid Winograd kernel (events around every call): median / max and the outliers, to tell a real stall of the
kernel from noise of the box.  Usage (GPU box): python tools/w42_jitter.py [--iters 60] [--op winograd42|winograd2d]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gddim_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--op", default="winograd42")
a = ap.parse_args()
dev = torch.device("cuda", 0)
fn_op = getattr(ops, "conv3x3_" + a.op)
for (H, Cin, Cout) in [(32, 128, 128), (32, 384, 128), (16, 256, 256), (16, 512, 256), (16, 384, 256), (8, 256, 256)]:
    B = 256
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, device=dev); r = torch.randn(B, H, H, Cout, device=dev); u = torch.empty(24 * Cout * Cin, device=dev)
    fn_op(x, w, b, r, out_scale=0.7, u_scratch=u)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
    for e0, e1 in ev:
        e0.record()
        fn_op(x, w, b, r, out_scale=0.7, u_scratch=u)
        e1.record()
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    raw = [e0.elapsed_time(e1) for e0, e1 in ev]
    out = [(i, round(v, 3)) for i, v in enumerate(raw) if v > 2 * t[len(t) // 2]]
    print("H=%2d %3d->%3d  median %.3f ms  min %.3f  max %.3f  outliers(>2x median) %s" % (H, Cin, Cout, t[len(t) // 2], t[0], t[-1], out), flush=True)
