"""What the reference's device axis costs on the host side: psampler with n_dev = 2 engines on the ONE GPU of a test box
(GDDIM_LOCAL_DEVICES=0,0: two engines, two streams, two whole-call hipGraphs in flight at once) against one engine with the same total batch,
CLD gDDIM order 2, NFE 50, full accr network.  Two engines on one GPU share its CUs, so the device time cannot drop; what this shows is that
the per-device calls cost the host (hipGraphLaunch of a 21 350-node graph is not free) with one issuing thread per device
(the default, gddim_amd/devices.py) and with the calls back to back from one thread (GDDIM_DEVICE_THREADS=0).  python tools/devices_overhead.py [per-device batch, default 128]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

Bd = int(sys.argv[1]) if len(sys.argv) > 1 else 128
from gddim_amd import devices  # noqa: E402
from gddim_amd.cld import sampling, sde_lib  # noqa: E402
from gddim_amd.configs import cld_accr_dcifar10  # noqa: E402
from gddim_amd.datasets import get_data_inverse_scaler  # noqa: E402
from gddim_amd.models import utils as mutils  # noqa: E402


def run(dev_list, B, threads=1, reps=3):
    os.environ["GDDIM_LOCAL_DEVICES"] = dev_list
    os.environ["GDDIM_DEVICE_THREADS"] = str(threads)
    n_dev = devices.local_device_count()
    cfg = cld_accr_dcifar10.get_config()
    cfg.sampling.method, cfg.sampling.nfe, cfg.sampling.deis_order = "deis", 50, 2
    model, _, params = mutils.init_model(42, cfg, init_scale_override=1.0)
    state = mutils.State(params_ema=params, model_state={})
    sde = sde_lib.from_config(cfg)
    fn = sampling.get_sampling_fn(cfg, sde, model, None, get_data_inverse_scaler(cfg))
    g = torch.Generator(device="cuda:0").manual_seed(1)
    u = torch.stack([torch.randn((n_dev, B, 32, 32, 3), generator=g, device="cuda:0"), torch.randn((n_dev, B, 32, 32, 3), generator=g, device="cuda:0") / 2], -1)
    seeds = list(range(n_dev))
    fn(seeds, state, B, u)
    torch.cuda.synchronize()
    ts, issue = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        xs, vs, n = fn(seeds, state, B, u)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        issue.append(t1 - t0)
    model.close()
    return {"n_dev": n_dev, "batch_per_device": B, "s_per_call_median": float(np.median(ts)), "images_per_s": n_dev * B / float(np.median(ts)),
            "host_issue_ms_median": 1e3 * float(np.median(issue))}


out = {"two_engines_one_gpu_threaded_issue": run("0,0", Bd, 1), "two_engines_one_gpu_serial_issue": run("0,0", Bd, 0), "one_engine": run("0", 2 * Bd)}
out["ratio_two_threaded_over_one"] = out["two_engines_one_gpu_threaded_issue"]["images_per_s"] / out["one_engine"]["images_per_s"]
out["ratio_two_serial_over_one"] = out["two_engines_one_gpu_serial_issue"]["images_per_s"] / out["one_engine"]["images_per_s"]
print(json.dumps(out, indent=1))
