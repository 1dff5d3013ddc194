"""One process per GPU: batch sharding and the end-of-round all-gather.

The reference shards a sampling round with jax.pmap (no collective inside the sampler,
cld_jax/sampling.py:236) and concatenates on the host (cld_jax/run_lib.py:722-725).  Here every rank
owns one MI355X and an identical engine; the only exchange is one all-gather of the uint8 images at
the end of a round (RCCL over xGMI with backend 'nccl'; 'gloo' on CPU for tests)."""
import os

import numpy as np


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_batch(global_batch, world, rank):
    """Contiguous split of a round's images; earlier ranks take the remainder (sizes differ by at most 1)."""
    base, rem = divmod(int(global_batch), int(world))
    size = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, size


def rank_seed(seed, round_index, rank):
    """Per-(round, rank) seed for the prior draw; the reference splits one PRNG key per round
    (cld_jax/run_lib.py:715) -- any injective mixing gives independent streams."""
    return (int(seed) * 1000003 + int(round_index)) * 65537 + int(rank)


def to_uint8(x):
    """np.clip(x * 255, 0, 255).astype(np.uint8) (cld_jax/run_lib.py:723): truncation, not rounding."""
    import torch
    if hasattr(x, 'is_cuda'):
        return torch.clamp(x * 255.0, 0.0, 255.0).to(torch.uint8)
    return np.clip(np.asarray(x) * 255.0, 0, 255).astype(np.uint8)


def all_gather_samples(local_u8, world=None):
    """Gather every rank's uint8 images (B_r,H,W,C) into (sum B_r,H,W,C) on every rank.  Shards may differ
    in size by one image; they are padded to the largest shard for the collective and trimmed after."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_u8
    world = dist.get_world_size()
    n = torch.tensor([local_u8.shape[0]], dtype=torch.int64, device=local_u8.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = local_u8
    if local_u8.shape[0] < mx:
        pad = torch.cat([local_u8, local_u8.new_zeros((mx - local_u8.shape[0],) + tuple(local_u8.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous())
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])
