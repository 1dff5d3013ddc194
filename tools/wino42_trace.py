"""In-kernel timeline of the persistent hybrid Winograd kernel: build a trace copy of the library
  python tools/build_variant.py w42trace kernels/conv3x3_wino42_f32.hip -DW42_TRACE
and run  GDDIM_LIB=gddim_amd/libgddim_w42trace.so python tools/wino42_trace.py  on the GPU box.  Thread 0 of
workgroups 0 and 100 stamps the 100 MHz wall clock and the shader clock into the tail of the residual tensor: at the top of a tile, after
its first chunk, after its K loop, after the exchange's barrier,
after the gather's barrier and after the stores were issued (-DW42_TRACE=2 and W42_TRACE_FINE=1: also after a wait for everything in flight
at the top of the epilogue and after the bias has arrived)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gddim_amd import ops
dev = torch.device("cuda", 0)
fine = os.environ.get("W42_TRACE_FINE", "0") == "1"  # a library built with -DW42_TRACE=2
per = 8 if fine else 6
names = ["first chunk", "rest of K loop", "x transform + exchange write", "gather"] + (["wait for the loads in flight", "bias", "arithmetic + stores"] if fine else ["epilogue + stores"])
for (H, Cin, Cout) in [(32, 128, 128), (32, 256, 128), (16, 256, 256), (16, 512, 256)]:
    B = 256
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, device=dev); r = torch.randn(B, H, H, Cout, device=dev); u = torch.empty(24 * Cout * Cin, device=dev)
    for _ in range(3):
        ops.conv3x3_winograd42(x, w, b, r, out_scale=0.7, u_scratch=u)
    torch.cuda.synchronize()
    t = r.flatten()[-128:].cpu().numpy()
    for slot, name in enumerate(("wg0", "wg100")):
        wall = t[slot * 64: slot * 64 + 24] / 100.0  # us
        cyc = t[slot * 64 + 32: slot * 64 + 56]
        for tile in range(24 // per):
            w5, c5 = wall[tile * per: tile * per + per], cyc[tile * per: tile * per + per]
            d = [(w5[i + 1] - w5[i]) for i in range(per - 1)]
            nch = Cin // 16
            dc = (c5[2] - c5[1]) % (1 << 24)
            nxt = (wall[tile * per + per] - w5[0]) if (tile + 1) * per < 24 else float("nan")
            print("H=%d %d->%d %-5s tile %d: " % (H, Cin, Cout, name, tile) + "  ".join("%s %.2f" % (names[i], d[i]) for i in range(per - 1)) +
                  " | per later chunk %.3f us, %.0f cycles, clock %.3f GHz | tile to tile %.2f us"
                  % (d[1] / (nch - 1), dc / (nch - 1), dc / d[1] / 1e3 if d[1] > 0 else 0, nxt))
