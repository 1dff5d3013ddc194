"""The persistent short-K implicit-GEMM kernel alone on the 1x1 / NIN shapes of the U-Net at B = 256 (bias, with / without residual), same box: profiles/r04_notes.md."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from gddim_amd import ops
dev = torch.device("cuda", 0)
B = 256
for H, Cout in ((32, 128), (16, 256)):
    for Cin in (128, 256, 384, 512, 768, 1024):
        x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev) / Cin ** 0.5; b = torch.randn(Cout, device=dev)
        r = torch.randn(B, H, H, Cout, device=dev)
        for res in (None, r):
            for _ in range(3): ops.conv2d(x, w, b, res, KH=1, KW=1, stride=1, pad=0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.conv2d(x, w, b, res, KH=1, KW=1, stride=1, pad=0)
            e1.record(); torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / 20
            fl = 2.0 * B * H * H * Cout * Cin
            print("1x1 H=%d %4d->%3d res=%d  %8.1f us  %6.1f TFLOP/s" % (H, Cin, Cout, res is not None, us, fl / us / 1e6), flush=True)
