"""In-kernel timeline of the 2-D Winograd kernel: build a trace copy of the library
  python tools/build_variant.py w2trace kernels/conv3x3_wino2d_f32.hip -DW2_TRACE
and run  GDDIM_LIB=gddim_amd/libgddim_w2trace.so python tools/wino2d_trace.py  on the GPU box.  Thread 0 of workgroups 0, 700, 1500 and 2040
stamps the 100 MHz wall clock at entry, after the prologue, after the K loop, after the LDS exchange, after the stores were issued and
after they drained, into the tail of the residual tensor."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gddim_amd import ops
dev = torch.device("cuda", 0)
for (H, Cin, Cout) in [(32, 128, 128), (32, 256, 128), (16, 256, 256), (16, 512, 256)]:
    B = 256
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, device=dev); r = torch.randn(B, H, H, Cout, device=dev); u = torch.empty(Cout * 16 * Cin, device=dev)
    for _ in range(3):
        ops.conv3x3_winograd2d(x, w, b, r, out_scale=0.7, u_scratch=u)
    torch.cuda.synchronize()
    t = r.flatten()[-64:].cpu().numpy()
    base = t[0]
    for slot, name in enumerate(("wg0", "wg700", "wg1500", "wg2040")):
        st = t[slot * 16: slot * 16 + 6] / 100.0  # us
        d = [st[i + 1] - st[i] for i in range(5)]
        cy = t[slot * 16 + 8: slot * 16 + 14]
        dcy = (cy[2] - cy[1]) % (1 << 24)
        print("H=%d %d->%d %-7s start %+8.2f us | prologue %.2f  K loop %.2f (%.3f/chunk)  exchange %.2f  residual+stores issued %.2f  drain %.2f | total %.2f | K loop %.0f cycles/chunk, clock %.3f GHz"
              % (H, Cin, Cout, name, st[0] - base / 100.0, d[0], d[1], d[1] / (Cin // 8), d[2], d[3], d[4], st[5] - st[0], dcy / (Cin // 8), dcy / d[1] / 1e3 if d[1] > 0 else 0))
