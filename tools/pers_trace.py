"""Per-workgroup timeline of igemm_f32_pers_kernel (two workgroups per CU): build with
  python tools/build_variant.py perstrace kernels/igemm.hip -DIGEMM_TRACE=1
and run  GDDIM_LIB=gddim_amd/libgddim_perstrace.so python tools/pers_trace.py  on the GPU box.  Thread 0 of every workgroup stamps the
100 MHz wall clock at entry and exit; the question is how far apart the two co-resident workgroups of a CU finish (the older one wins the
matrix pipe's arbitration), i.e. what a dynamic tile queue instead of the static stride walk could recover."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import torch
from gddim_amd import ops
dev = torch.device("cuda", 0)
B = 256
for H, Cin, Cout in ((32, 256, 128), (32, 384, 128), (16, 512, 256), (16, 256, 256)):
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev) / Cin ** 0.5; b = torch.randn(Cout, device=dev)
    res = torch.zeros(B, H, H, Cout, device=dev)
    for _ in range(3):
        out = ops.conv2d(x, w, b, res, KH=1, KW=1, stride=1, pad=0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = ops.conv2d(x, w, b, res, KH=1, KW=1, stride=1, pad=0); e1.record(); torch.cuda.synchronize()
    G = 512
    tr = res.reshape(-1)[-3 * G:].cpu().numpy().astype(np.float64).reshape(G, 3)
    t0, t1, nt = tr[:, 0], tr[:, 1], tr[:, 2]
    base = t0.min()
    start, end = (t0 - base) / 100.0, (t1 - base) / 100.0  # us
    old, young = slice(0, G // 2), slice(G // 2, G)
    print("H=%d %d->%d: kernel %.1f us | tiles per workgroup %s | start us: first half %.1f..%.1f second half %.1f..%.1f | end us: first half %.1f +- %.1f (max %.1f), second half %.1f +- %.1f (max %.1f) | makespan %.1f, mean busy %.1f"
          % (H, Cin, Cout, 1e3 * e0.elapsed_time(e1), sorted(set(nt.astype(int).tolist())), start[old].min(), start[old].max(), start[young].min(), start[young].max(),
             end[old].mean(), end[old].std(), end[old].max(), end[young].mean(), end[young].std(), end[young].max(), end.max(), (end - start).mean()), flush=True)
    # per CU pairing is unknown to the program; the distribution of end times is what matters
    hist, edges = np.histogram(end, bins=10)
    print("   end-time histogram (us):", " ".join("%.0f:%d" % (edges[i], hist[i]) for i in range(10)))
