import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gddim_amd import ops
H, Cin, Cout = [int(a) for a in sys.argv[1:4]]
B = 256
dev = torch.device("cuda", 0)
x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5
b = torch.randn(Cout, device=dev); r = torch.randn(B, H, H, Cout, device=dev); u = torch.empty(Cout * 12 * Cin, device=dev)
for _ in range(5):
    ops.conv3x3_winograd(x, w, b, r, out_scale=0.7, u_scratch=u)
    ops.conv2d(x, w, b, r, out_scale=0.7)
torch.cuda.synchronize()
