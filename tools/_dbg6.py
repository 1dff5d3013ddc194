import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from gddim_amd import ops
import sys
B,H,W,C,N = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (1,16,16,16,64)
g = torch.Generator().manual_seed(1)
x = torch.randn((B,H,W,C), generator=g); w = torch.randn((N, 9*C), generator=g) * 0.1
wd = w.double().reshape(N,3,3,C).permute(0,3,1,2)
want = torch.nn.functional.conv2d(x.double().permute(0,3,1,2), wd, padding=1).permute(0,2,3,1)
u = ops.conv3x3_winograd2d_split6_weights(w.cuda())
got = ops.conv3x3_winograd2d_split6(x.cuda(), u, N).cpu().double()
e = (got - want).abs()
bad = e > 1e-4
print("bad fraction", bad.float().mean().item())
print("bad per output-row parity a:", [bad[0, a::2].float().mean().item() for a in range(2)])
print("bad per output-col parity b:", [bad[0, :, b::2].float().mean().item() for b in range(2)])
print("bad per tile row ty:", [round(bad[0, 2*t:2*t+2].float().mean().item(), 2) for t in range(8)])
print("bad per tile col tx:", [round(bad[0, :, 2*t:2*t+2].float().mean().item(), 2) for t in range(8)])
print("bad per n quad:", [round(bad[0, :, :, 4*q:4*q+4].float().mean().item(), 2) for q in range(16)])
print("bad per n mod 4:", [round(bad[0, :, :, k::4].float().mean().item(), 2) for k in range(4)])
idx = bad.nonzero()[:6]
for (b_, y, x_, n) in idx.tolist():
    print("pos", (y, x_, n), "got %.5f want %.5f" % (got[b_, y, x_, n], want[b_, y, x_, n]), " neighbours got", [round(float(got[b_, y, x_, n + d]), 4) for d in (-3, -2, -1)],
          "want", [round(float(want[b_, y, x_, n + d]), 4) for d in (-3, -2, -1)])
    # candidates: value at same n of other pixels of the tile
    ty, tx = y // 2, x_ // 2
    print("   tile pixels want:", [[round(float(want[b_, 2 * ty + a, 2 * tx + c, n]), 4) for c in range(2)] for a in range(2)], " n-16/+... got-want", round(float(got[b_, y, x_, n] - want[b_, y, x_, n]), 5))
