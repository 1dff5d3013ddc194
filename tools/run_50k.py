"""The north-star's own target run on ONE GPU: >= 50k samples at NFE = 50 through the reference's caller.

`run_lib.sample_data` (cld_jax/run_lib.py:674-731) with the reference's `eval.batch_size = 1024`, `eval.num_samples = 50000`
(49 rounds = 50 176 images), CLD gDDIM `deis_order = 2`, NFE = 50, the full accr NCSN++ with synthetic weights (no checkpoint
exists here; throughput does not depend on the weight values), one `samples_<r>.npz` per round with the reference's keys.

Writes one JSON object: wall time, sustained images/s INCLUDING the float32 gather, the device-to-host copies and
`np.savez_compressed`, per-round device-time spread, peak HBM in use (device-wide: the engine's arena and weights are hipMalloc'd
outside torch's allocator, so the figure is total - min(free)), peak host RSS, bytes on disk -- for the pipelined writer (default)
and, with --serial-rounds N, for N rounds of the reference's serial loop (writer_threads = 0) next to it.

    python tools/run_50k.py --out gpurun_out/r05_50k_run.json [--rounds 49] [--batch 1024] [--serial-rounds 4]
"""
import argparse
import json
import os
import resource
import shutil
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _HbmWatch(threading.Thread):
    """Samples the device's free memory twice a second (hipMemGetInfo through torch): peak in use = total - min(free)."""

    def __init__(self):
        super().__init__(daemon=True)
        import torch
        self._torch = torch
        self.min_free, self.total = torch.cuda.mem_get_info()
        self._stop_evt = threading.Event()

    def run(self):
        while not self._stop_evt.wait(0.5):
            free, _ = self._torch.cuda.mem_get_info()
            self.min_free = min(self.min_free, free)

    def stop(self):
        self._stop_evt.set()
        self.join()
        return self.total - self.min_free


def _run(cfg, state, folder, writers, rounds):
    import numpy as np
    from gddim_amd import run_lib
    cfg.eval.num_samples = (rounds - 1) * cfg.eval.batch_size + cfg.eval.batch_size // 2  # num_samples // batch + 1 == rounds
    cfg.eval.writer_threads = writers
    shutil.rmtree(folder, ignore_errors=True)
    watch = _HbmWatch()
    watch.start()
    stats = {}
    t0 = time.perf_counter()
    files = run_lib.sample_data(cfg, "unused", folder, state=state, stats=stats)
    wall = time.perf_counter() - t0
    peak_hbm = watch.stop()
    assert len(files) == rounds, (len(files), rounds)
    gpu_ms = np.asarray(stats.get("gpu_round_ms", [0.0]))
    info = np.load(files[-1])
    assert info["samples"].shape == (cfg.eval.batch_size, 32, 32, 3) and info["samples"].dtype == np.uint8
    assert set(info.files) == {"samples", "nfe_cnt", "samples_v", "samples_x"} and int(info["nfe_cnt"]) == cfg.sampling.nfe
    assert np.isfinite(info["samples_x"]).all()
    return {
        "writer_threads": writers, "rounds": rounds, "images": rounds * cfg.eval.batch_size,
        "wall_s_including_factory": round(wall, 3),                       # tables, engine creation, graph capture, every round, every file
        "wall_s_rounds": round(stats["wall_s"], 3),                       # first round submitted -> last file on disk
        "images_per_s_sustained": round(rounds * cfg.eval.batch_size / stats["wall_s"], 2),
        "gpu_round_ms": {"median": round(float(np.median(gpu_ms)), 1), "min": round(float(gpu_ms.min()), 1), "max": round(float(gpu_ms.max()), 1),
                         "p95": round(float(np.percentile(gpu_ms, 95)), 1)},
        "images_per_s_device_median_round": round(cfg.eval.batch_size / (float(np.median(gpu_ms)) / 1e3), 2) if gpu_ms.max() > 0 else None,
        "peak_hbm_in_use_gb": round(peak_hbm / 1e9, 2),
        "bytes_on_disk": sum(os.path.getsize(f) for f in files),
        "first_file_keys": sorted(info.files),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r05_50k_run.json")
    ap.add_argument("--rounds", type=int, default=49)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--nfe", type=int, default=50)
    ap.add_argument("--writers", type=int, default=2)
    ap.add_argument("--serial-rounds", type=int, default=4)
    ap.add_argument("--folder", default="/tmp/gddim_50k")
    args = ap.parse_args()
    import numpy as np
    import torch
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.models import utils as mutils
    cfg = cld_accr_dcifar10.get_config()
    cfg.eval.batch_size = args.batch
    cfg.sampling.method, cfg.sampling.nfe, cfg.sampling.deis_order, cfg.sampling.noise_removal = "deis", args.nfe, 2, True
    _, _, params = mutils.init_model(42, cfg, init_scale_override=1.0)  # the weights bench.py uses
    state = mutils.State(params_ema=params, model_state={})
    rec = {"what": "run_lib.sample_data, CLD gDDIM deis_order=2, NFE=%d, eval.batch_size=%d, accr NCSN++ (synthetic weights), 1 x MI355X, fp32" % (args.nfe, args.batch),
           "reference": "cld_jax/run_lib.py:674-731", "host_cores": os.cpu_count()}
    if args.serial_rounds > 0:
        rec["serial_loop"] = _run(cfg, state, args.folder + "_serial", 0, args.serial_rounds)
        shutil.rmtree(args.folder + "_serial", ignore_errors=True)
    rec["pipelined"] = _run(cfg, state, args.folder, args.writers, args.rounds)
    rec["peak_host_rss_gb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024 / 1e9, 2)
    rec["torch_max_memory_allocated_gb"] = round(torch.cuda.max_memory_allocated() / 1e9, 3)
    shutil.rmtree(args.folder, ignore_errors=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
