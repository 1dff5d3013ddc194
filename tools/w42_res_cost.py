"""What the residual read costs the hybrid Winograd kernel: the production launches with and without a residual tensor, same box; profiles/r05_notes.md."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from gddim_amd import ops
dev = torch.device("cuda", 0)
B = 256
for (H, Cin, Cout) in [(32, 128, 128), (32, 256, 128), (16, 256, 256), (16, 512, 256), (8, 256, 256)]:
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, device=dev); r = torch.randn(B, H, H, Cout, device=dev); u = torch.empty(24 * Cout * Cin, device=dev)
    out = {}
    for tag, bb, rr in (("bias+res", b, r), ("bias", b, None), ("none", None, None), ("bias+res", b, r), ("bias", b, None)):
        for _ in range(3): ops.conv3x3_winograd42(x, w, bb, rr, out_scale=0.7, u_scratch=u)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv3x3_winograd42(x, w, bb, rr, out_scale=0.7, u_scratch=u)
        e1.record(); torch.cuda.synchronize()
        out.setdefault(tag, []).append(1e3 * e0.elapsed_time(e1) / 20)
    print("H=%d %d->%d (incl. the weight transform): " % (H, Cin, Cout) + "  ".join("%s %s us" % (k, "/".join("%.1f" % v for v in vs)) for k, vs in out.items()), flush=True)
