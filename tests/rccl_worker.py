"""Worker of tests/test_gpu_rccl.py: one rank of a `torch.distributed.run` launch on a GPU box.  Builds the `nccl` (= RCCL on
ROCm) process group and pushes the end-of-round gather of gddim_amd.distributed through it on HBM-resident uint8 images --
both the equal-shard path (one all_gather_into_tensor) and the general path (size exchange + padded all_gather) -- forcing the
collective even though the group has one rank.  Prints RCCL_OK on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from gddim_amd import distributed  # noqa: E402


def main():
    rank, world, local_rank = distributed.init_from_env(backend="nccl", single_rank_group=True)
    assert dist.is_initialized() and dist.get_backend() == "nccl", "RCCL group not initialised"
    dev = torch.device("cuda", local_rank)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    x = torch.rand((256, 32, 32, 3), generator=g, device=dev) * 1.2 - 0.1  # one round of sampler output, in and out of [0, 1]
    u8 = distributed.to_uint8(x)
    assert u8.is_cuda and u8.dtype == torch.uint8
    a = distributed.all_gather_samples(u8, equal_shards=True, force=True)
    b = distributed.all_gather_samples(u8, force=True)
    torch.cuda.synchronize()
    start = rank * 256
    assert a.shape == (256 * world, 32, 32, 3) and torch.equal(a, b), "the two gather paths disagree"
    assert torch.equal(a[start:start + 256], u8), "gathered shard differs from the local one"
    fl = distributed.all_gather_samples(x, equal_shards=True, force=True)  # the float copies run_lib also gathers
    assert torch.equal(fl[start:start + 256], x)
    dist.barrier()
    if rank == 0:
        try:
            ver = str(torch.cuda.nccl.version())
        except Exception as exc:  # version query is informational only
            ver = "unknown (%s)" % type(exc).__name__
        print("RCCL_OK backend=%s world=%d rccl_version=%s bytes=%d" % (dist.get_backend(), world, ver, a.numel()))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
