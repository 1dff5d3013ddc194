"""Per layer class: the 3x3 convolutions of one U-Net evaluation at B = 256 in fp32 mode (hybrid Winograd kernel, epilogue GroupNorm where it
fuses) and in split6 mode, with the GroupNorm passes that follow them -- which classes the split6 kernel should take.
GD_PROF_BY_LABEL=2 python tools/split6_layers.py"""
import os, sys, json, collections
os.environ["GD_PROF_BY_LABEL"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gddim_amd.configs import cld_accr_dcifar10
from gddim_amd.engine import Engine
from gddim_amd.models import utils as mutils
cfg = cld_accr_dcifar10.get_config()
B = 256
_, _, params = mutils.init_model(42, cfg, init_scale_override=1.0)
eng = Engine(cfg, 6, max_batch=B, device=0).load_params(params)
x = torch.randn(B, 32, 32, 6, device="cuda:0"); lab = torch.full((B,), 500.0, device="cuda:0")
res = {}
for mode in ("fp32", "split6"):
    eng.set_precision(mode)
    for _ in range(2):
        eng.unet(x, lab)
    tot = collections.defaultdict(lambda: [0.0, 0])
    for rep in range(5):
        eng.profile(True); eng.unet(x, lab); ents = eng.profile_read(); eng.profile(False)
        for p in ents:
            tot[p["name"]][0] += p["ms"] / 5; tot[p["name"]][1] = p["launches"]
    res[mode] = dict(tot)
    print(mode, "evaluation total %.2f ms" % sum(v[0] for v in tot.values()))
def cls(name):  # "label class|kernel" -> (class, kernel)
    a, _, b = name.partition("|")
    return a, b
rows = collections.defaultdict(dict)
for mode, d in res.items():
    for name, (ms, n) in d.items():
        c, k = cls(name)
        rows[c][mode] = rows[c].get(mode, 0.0) + ms
        rows[c][mode + "_k"] = k
        rows[c][mode + "_n"] = n
print("%-44s %10s %10s   kernels" % ("class", "fp32 ms", "split6 ms"))
for c, r in sorted(rows.items(), key=lambda kv: -max(kv[1].get("fp32", 0), kv[1].get("split6", 0))):
    if max(r.get("fp32", 0), r.get("split6", 0)) < 0.05: continue
    print("%-44s %10.3f %10.3f   %s (%s) | %s" % (c, r.get("fp32", 0), r.get("split6", 0), r.get("fp32_k", "-"), r.get("fp32_n", "-"), r.get("split6_k", "-")))
eng.close()
