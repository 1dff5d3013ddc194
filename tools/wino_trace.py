"""In-kernel timeline of the persistent Winograd kernel: build a trace copy of the library
  (hipcc ... -DWINO_TRACE -c gddim_amd/csrc/kernels/conv3x3_wino_f32.hip; link with the other objects of gddim_amd/_build into another .so)
and run  GDDIM_LIB=<that .so> python tools/wino_trace.py  on the GPU box.  Workgroups 0 and 100 write (wall clock, shader cycle counter)
pairs at every second chunk and around the epilogue into the tail of the residual tensor; printed: time and cycles per two chunks and
the shader clock they imply."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gddim_amd import ops
dev = torch.device("cuda", 0)
for (H, Cin, Cout) in [(32, 128, 128), (32, 256, 128), (16, 512, 256)]:
    B = 256
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, device=dev); r = torch.randn(B, H, H, Cout, device=dev); u = torch.empty(Cout * 12 * Cin, device=dev)
    for _ in range(3):
        ops.conv3x3_winograd(x, w, b, r, out_scale=0.7, u_scratch=u)
    torch.cuda.synchronize()
    t = r.flatten()[-256:].cpu().numpy()
    for name, off in (("wg0", 0), ("wg100", 128)):
        import numpy as np
        wall = t[off:off + 120:2] / 100.0; cyc = t[off + 1:off + 120:2]
        dw, dc = np.diff(wall)[2:20], np.diff(cyc)[2:20]
        print(H, Cin, Cout, name, "us per 2 chunks %.2f  cycles per 2 chunks %.0f  -> clock %.3f GHz" % (dw.mean(), dc.mean(), dc.mean() / dw.mean() / 1e3))
