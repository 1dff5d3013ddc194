"""In-kernel timeline of the split6 prototype: build  python tools/build_variant.py s6trace kernels/conv3x3_wino2d_split6.hip -fno-slp-vectorize -DS6_TRACE
and run  GDDIM_LIB=gddim_amd/libgddim_s6trace.so python tools/split6_trace.py  on the GPU box."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gddim_amd import ops
dev = torch.device("cuda", 0)
for (H, C) in [(16, 256), (32, 128)]:
    B, N = 256, C
    x = torch.randn(B, H, H, C, device=dev); w = torch.randn(N, 9 * C, device=dev) / (9 * C) ** 0.5
    b = torch.randn(N, device=dev); r = torch.zeros(B, H, H, N, device=dev)
    u = ops.conv3x3_winograd2d_split6_weights(w)
    for _ in range(3):
        ops.conv3x3_winograd2d_split6(x, u, N, bias=b, res=r)
    torch.cuda.synchronize()
    t = r.flatten()[-1024:].cpu().numpy().astype(np.float64)
    NP = 2 * (C // 16)
    for slot, name in enumerate(("wg0 wave0", "wg0 wave4", "wg201 wave0", "wg201 wave4")):
        wall = t[slot * 160: slot * 160 + 80] / 100.0
        cyc = t[slot * 160 + 80: slot * 160 + 160]
        d = lambda a, i, j: (a[j] - a[i]) % (1 << 24 if a is cyc else (1 << 24) / 100.0)  # noqa: E731
        tot_us = d(wall, 0, 42)
        if tot_us <= 0:
            continue
        clock = d(cyc, 1, 2 + NP - 1) / d(wall, 1, 2 + NP - 1) / 1e3 if d(wall, 1, 2 + NP - 1) > 0 else 0
        phases = [d(cyc, 1 + i, 2 + i) for i in range(min(NP, 32))]
        print("H=%d C=%d %-12s total %.2f us: prologue %.2f, K loop %.2f (%.0f cycles per phase, median %.0f; clock %.2f GHz), tail MFMAs %.2f, z exchange %.2f, outputs %.2f"
              % (H, C, name, tot_us, d(wall, 0, 1), d(wall, 1, 2 + NP - 1), np.mean(phases), np.median(phases), clock, d(wall, 2 + NP - 1, 39),
                 d(wall, 39, 41), d(wall, 41, 42)))
        print("      phases (cycles): " + " ".join("%.0f" % v for v in phases))
