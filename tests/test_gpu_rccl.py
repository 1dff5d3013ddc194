"""RCCL on real hardware: the sampler's only collective (the end-of-round all-gather of uint8 images, gddim_amd.distributed) run
through the `nccl` backend of torch.distributed -- which is RCCL on ROCm -- in a 1-rank group on the GPU box, and the benchmark
launched the way the driver launches it for N > 1 (`python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`).
The N = 2 host logic is covered on CPU by tests/test_distributed_cpu.py (gloo)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(script_args, timeout):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_rccl_all_gather_one_rank(native_lib):
    r = _launch([os.path.join(ROOT, "tests", "rccl_worker.py")], 600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "RCCL_OK backend=nccl" in r.stdout


def test_bench_under_the_launcher_one_rank(native_lib):
    """bench.py exactly as the driver launches it for N > 1, at N = 1: group built, barrier + max-over-ranks timing, RCCL gather."""
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--batch", "32", "--nfe", "4",
                 "--no-cpu-baseline", "--no-alt", "--no-profile"], 900)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and "RCCL" in out["config"]["parallelism"]
