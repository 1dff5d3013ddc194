"""One process per GPU: batch sharding and the end-of-round all-gather.

The reference shards a sampling round with jax.pmap (no collective inside the sampler,
cld_jax/sampling.py:236) and concatenates on the host (cld_jax/run_lib.py:722-725).  Here every rank
owns one MI355X and an identical engine; the only exchange is one all-gather of the uint8 images at
the end of a round (RCCL over xGMI with backend 'nccl'; 'gloo' on CPU for tests)."""
import os

import numpy as np


def init_from_env(backend=None, single_rank_group=False):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank).
    single_rank_group=True also builds the (1-rank) group when the launcher set RANK with WORLD_SIZE=1, so that a
    `torch.distributed.run --nproc-per-node 1` launch runs its end-of-round gather through RCCL like the N>1 launches do."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if (world > 1 or (single_rank_group and launched)) and not dist.is_initialized():
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


def any_rank_failed(failed):
    """True on every rank as soon as one rank reports a failure: one 4-byte all-reduce, a no-op outside a process group.  run_lib asks
    it at the top of every round so that a rank whose file writer died does not leave the others waiting in the round's gather."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(failed)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    return bool(int(flag.item()))


def all_gather_samples(local, equal_shards=None, force=False):
    """Gather every rank's images (B_r, ...) into (sum B_r, ...) on every rank -- one collective per round, uint8 for the
    FID images (3 072 B per image), the only data-path exchange of the sampler.

    equal_shards=True (the caller split the round with `shard_batch` and the batch divides the world size): one
    `all_gather_into_tensor` straight into the output, no size exchange, no host synchronisation.  equal_shards=None/False:
    shard sizes (which differ by at most one image) are exchanged first, shards are padded to the largest one for the
    collective and trimmed after.  force=True runs the collective even in a 1-rank group (used to exercise RCCL on one GPU)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return local
    world = dist.get_world_size()
    local = local.contiguous()
    if equal_shards:
        out = local.new_empty((world * local.shape[0],) + tuple(local.shape[1:]))
        if dist.get_backend() == "gloo":  # gloo has no all_gather_into_tensor
            dist.all_gather(list(out.chunk(world)), local)
        else:
            dist.all_gather_into_tensor(out, local)
        return out
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s) for s in torch.cat(sizes).tolist()]
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous())
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])
