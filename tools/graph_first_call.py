"""First-call cost of the sampler's launch forms at one batch size (eager / per-step hipGraph / one graph per call): capture time and graph memory; profiles/r05_notes.md."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from gddim_amd.cld import sampling, sde_lib
from gddim_amd.configs import cld_accr_dcifar10
from gddim_amd.datasets import get_data_inverse_scaler
from gddim_amd.models import utils as mutils
dev = torch.device("cuda", 0)
cfg = cld_accr_dcifar10.get_config()
cfg.sampling.nfe, cfg.sampling.deis_order, cfg.sampling.method = 50, 2, "deis"
model, _, params = mutils.init_model(42, cfg, init_scale_override=1.0)
state = mutils.State(params_ema=params, model_state={})
sde = sde_lib.from_config(cfg)
fn = sampling.get_sampling_fn(cfg, sde, model, None, get_data_inverse_scaler(cfg))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
u = torch.randn((1, B, 32, 32, 3, 2), device=dev)
fn([0], state, B, u, use_graph=False); torch.cuda.synchronize()   # engine creation, weights, eager warm-up
free0 = torch.cuda.mem_get_info()[0]
for mode in (True, "loop"):
    t0 = time.perf_counter(); fn([0], state, B, u, use_graph=mode); torch.cuda.synchronize(); t1 = time.perf_counter()
    fn([0], state, B, u, use_graph=mode); torch.cuda.synchronize(); t2 = time.perf_counter()
    free1 = torch.cuda.mem_get_info()[0]
    print("B=%d use_graph=%r: first call %.3f s (capture + instantiate + run), second %.3f s, device memory taken by the graph(s) %.1f MB" % (B, mode, t1 - t0, t2 - t1, (free0 - free1) / 1e6), flush=True)
    free0 = free1
