"""One layer shape through the hybrid Winograd kernel a few times (PMC / trace runs): python tools/w42_one.py H Cin Cout [op]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gddim_amd import ops
H, Cin, Cout = [int(a) for a in sys.argv[1:4]]
op = getattr(ops, "conv3x3_" + (sys.argv[4] if len(sys.argv) > 4 else "winograd42"))
B = 256
dev = torch.device("cuda", 0)
x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5
b = torch.randn(Cout, device=dev); r = torch.randn(B, H, H, Cout, device=dev); u = torch.empty(24 * Cout * Cin, device=dev)
for _ in range(5):
    op(x, w, b, r, out_scale=0.7, u_scratch=u)
torch.cuda.synchronize()
