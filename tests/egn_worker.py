"""Worker of tests/test_gpu_unet.py::test_epilogue_groupnorm_16x16_level: one NCSN++ forward of a small three-level network
(32x32 / 16x16 / 8x8) on the GPU in a process of its own, because the switches it is run under (GDDIM_WINO2_MINBLOCKS=1: the
two-dimensional Winograd kernel at a small batch; GDDIM_NO_EGN=1: no GroupNorm in the producer's epilogue) are read once per process.
Writes the output, the float64 oracle's output and the per-kernel launch counts to the .npz path given as argv[1]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from conftest import tiny_config  # noqa: E402
from oracle import ncsnpp as onet  # noqa: E402


def main():
    from gddim_amd.engine import Engine
    nf = int(os.environ.get("EGN_WORKER_NF", "32"))  # 64: the 32x32 level has Cin = 64, the persistent form of the hybrid kernel
    cfg = tiny_config("cld", nf=nf, num_res_blocks=2, image_size=32, ch_mult=(1, 2, 2), attn=(16,))
    B, in_ch = 5, 6
    params = onet.random_params(cfg, in_ch, seed=21)
    eng = Engine(cfg, in_ch, max_batch=B, device=0)
    eng.load_params(params)
    rng = np.random.default_rng(22)
    x = rng.standard_normal((B, 32, 32, in_ch)).astype(np.float32)
    labels = (rng.uniform(1e-3, 1.0, size=B) * 999).astype(np.float32)
    eng.profile(True)
    got = eng.unet(torch.as_tensor(x, device="cuda:0"), torch.as_tensor(labels, device="cuda:0")).cpu().numpy()
    torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile(False)
    with torch.no_grad():
        want = onet.ncsnpp_forward(params, torch.as_tensor(x, dtype=torch.float64), torch.as_tensor(labels, dtype=torch.float64), cfg).numpy()
    eng.close()
    np.savez(sys.argv[1], got=got, want=want, names=np.array([p["name"] for p in prof]), launches=np.array([p["launches"] for p in prof]))
    print("EGN_WORKER_OK")


if __name__ == "__main__":
    main()
