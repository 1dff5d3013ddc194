"""In-kernel timeline of the persistent hybrid Winograd kernel: build a trace copy of the library
  python tools/build_variant.py w42trace kernels/conv3x3_wino42_f32.hip -DW42_TRACE
and run  GDDIM_LIB=gddim_amd/libgddim_w42trace.so python tools/wino42_trace.py  on the GPU box.  Thread 0 of workgroups 0 and 100 stamps the
100 MHz wall clock and the shader clock at the top of a tile, after its K loop, after the first exchange barrier, after the second one
and after the stores were issued, for its first four tiles, into the tail of the residual tensor."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gddim_amd import ops
dev = torch.device("cuda", 0)
for (H, Cin, Cout) in [(32, 128, 128), (32, 256, 128), (16, 256, 256), (16, 512, 256)]:
    B = 256
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, device=dev); r = torch.randn(B, H, H, Cout, device=dev); u = torch.empty(24 * Cout * Cin, device=dev)
    for _ in range(3):
        ops.conv3x3_winograd42(x, w, b, r, out_scale=0.7, u_scratch=u)
    torch.cuda.synchronize()
    t = r.flatten()[-128:].cpu().numpy()
    for slot, name in enumerate(("wg0", "wg100")):
        wall = t[slot * 64: slot * 64 + 20] / 100.0  # us
        cyc = t[slot * 64 + 32: slot * 64 + 52]
        for tile in range(4):
            w5, c5 = wall[tile * 5: tile * 5 + 5], cyc[tile * 5: tile * 5 + 5]
            d = [(w5[i + 1] - w5[i]) for i in range(4)]
            dc = (c5[1] - c5[0]) % (1 << 24)
            nxt = (wall[tile * 5 + 5] - w5[4]) if tile < 3 else float("nan")
            print("H=%d %d->%d %-5s tile %d: K loop %.2f us (%.3f per 16-ch chunk, %.0f cycles/chunk, clock %.3f GHz)  x-transform+exchange write %.2f  "
                  "gather %.2f  epilogue+stores %.2f | tile %.2f us, to next tile %.2f"
                  % (H, Cin, Cout, name, tile, d[0], d[0] / (Cin // 16), dc / (Cin // 16), dc / d[0] / 1e3 if d[0] > 0 else 0, d[1], d[2], d[3],
                     w5[4] - w5[0], nxt))
