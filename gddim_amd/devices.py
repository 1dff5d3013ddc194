"""The device axis of the reference's `psampler` (cld_jax/sampling.py:232-237, blur_jax/sampling.py:82-87) without jax.pmap.

In the reference ONE process drives every local accelerator: `prng` is a stack of `jax.local_device_count()` keys, `u` is
`(n_dev, B, H, W, C[, 2])`, the result is a sharded array of shape `(n_dev, B, H, W, C)` whose shards stay on their devices
until the caller's `np.clip(samples_org_x * 255., 0, 255)` pulls them to the host (cld_jax/run_lib.py:711-725).  Here the same
thing is `n_dev` native engines, one per local GPU: a sampler call is ONE asynchronous `hipGraphLaunch` per engine (the whole-call
graph), each issued by its own short-lived task on a per-device thread (`_issue_threaded`: the launch of a
21 350-node graph costs the host 185-360 ms, so one thread issuing them back to back would start device d that much late per
device before it), so the GPUs run concurrently while the caller has already returned.

  local_devices()        the HIP device indices this process drives
  local_device_count()   jax.local_device_count()
  ShardedDeviceArray     the (n_dev, B, ...) result: per-device torch tensors, host conversion on demand
  pmap_sampler(...)      pmap(sampler, static_broadcasted_argnums=2)(prng, pstate, batch_size, u)

Which devices: `GDDIM_LOCAL_DEVICES` (comma list of HIP device indices; an index may repeat -- two engines on one GPU, what the
one-GPU test box uses to exercise n_dev = 2) > one device, LOCAL_RANK, when a launcher started one process per GPU (WORLD_SIZE >
1: gddim_amd.distributed, bench.py --gpus N) > every visible GPU."""
import os

import numpy as np

from ._abi import GddimError


def local_devices():
    import torch
    spec = os.environ.get("GDDIM_LOCAL_DEVICES", "").strip()
    if spec:
        try:
            devs = [int(s) for s in spec.split(",") if s.strip() != ""]
        except ValueError:
            raise GddimError("GDDIM_LOCAL_DEVICES must be a comma list of device indices (got %r)" % spec)
        if not devs or min(devs) < 0:
            raise GddimError("GDDIM_LOCAL_DEVICES must name at least one non-negative device index (got %r)" % spec)
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n and max(devs) >= n:
            raise GddimError("GDDIM_LOCAL_DEVICES=%s names a device this process cannot see (%d visible)" % (spec, n))
        return devs
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:  # one process per GPU: this process owns LOCAL_RANK
        if torch.cuda.is_available():
            return [torch.cuda.current_device()]
        return [int(os.environ.get("LOCAL_RANK", "0"))]
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    return list(range(n)) if n else [0]


def local_device_count():
    return len(local_devices())


def device_seeds(prng, n_dev):
    """One integer seed per device from the reference's stack of per-device PRNG keys (cld_jax/run_lib.py:715-720:
    `jax.random.split(rng, n_dev + 1)[1:]`).  A stack with one entry per device gives every device its own; a shorter stack
    (a bare seed, `[seed]`) is spread with an injective mix so that no two devices draw the same stream."""
    from .cld.sampling import _seed_of
    keys = None
    if prng is not None and not isinstance(prng, (int, np.integer)) and not hasattr(prng, 'integers'):
        try:
            if len(prng) == n_dev and n_dev > 1:
                keys = [prng[d] for d in range(n_dev)]
        except TypeError:
            keys = None
    if keys is not None:
        return [_seed_of(k) for k in keys]
    try:
        first = prng[0] if not isinstance(prng, (int, np.integer)) and not hasattr(prng, 'integers') and prng is not None else prng
    except (TypeError, IndexError):
        first = prng
    base = _seed_of(first)
    return [base if d == 0 else (base * 6364136223846793005 + d * 1442695040888963407) % (1 << 62) for d in range(n_dev)]


class ShardedDeviceArray(object):
    """`(n_dev, B, ...)` array whose slice d lives on device d (what jax.pmap returns).  Index with an int for the device tensor of
    a shard; anything that needs the whole array (`np.asarray`, `np.clip(a * 255., 0, 255)`, `np.savez_compressed(..., a=a)`,
    `.cpu()`, `.numpy()`, `.reshape`) concatenates on the host, exactly where the reference does (cld_jax/run_lib.py:723).
    Arithmetic with scalars or with another sharded array of the same layout stays on the devices."""

    __array_priority__ = 100.0

    def __init__(self, shards):
        shards = list(shards)
        if not shards:
            raise ValueError("a sharded array needs at least one shard")
        s0 = tuple(shards[0].shape)
        for s in shards:
            if tuple(s.shape) != s0:
                raise ValueError("shards differ in shape: %r vs %r" % (tuple(s.shape), s0))
        self.shards = shards

    # -- layout --
    @property
    def shape(self):
        return (len(self.shards),) + tuple(self.shards[0].shape)

    @property
    def ndim(self):
        return 1 + self.shards[0].dim()

    @property
    def dtype(self):
        return self.shards[0].dtype

    @property
    def devices(self):
        return [s.device for s in self.shards]

    def __len__(self):
        return len(self.shards)

    def __iter__(self):
        return iter(self.shards)

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            return self.shards[int(idx)]
        if isinstance(idx, tuple) and idx and isinstance(idx[0], (int, np.integer)):
            return self.shards[int(idx[0])][idx[1:]]
        return self.numpy()[idx]

    def __repr__(self):
        return "ShardedDeviceArray(shape=%r, dtype=%s, devices=%s)" % (self.shape, self.dtype, [str(d) for d in self.devices])

    # -- host side --
    def block_until_ready(self):
        import torch
        for s in self.shards:
            if s.is_cuda:
                torch.cuda.synchronize(s.device)
        return self

    def cpu(self):
        """One host tensor (n_dev, B, ...): every shard's copy is queued before the first one is waited for."""
        import torch
        if len(self.shards) == 1:
            return self.shards[0].detach().cpu()[None]
        out = torch.empty(self.shape, dtype=self.dtype, pin_memory=any(s.is_cuda for s in self.shards))
        for d, s in enumerate(self.shards):
            out[d].copy_(s.detach(), non_blocking=True)
        self.block_until_ready()
        return out

    def numpy(self):
        return self.cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)

    def reshape(self, *shape):
        return self.numpy().reshape(*shape)

    def astype(self, dtype):
        return self.numpy().astype(dtype)

    # -- device side --
    def _map(self, fn, other=None):
        if isinstance(other, ShardedDeviceArray):
            if other.shape != self.shape:
                raise ValueError("sharded operands differ in shape: %r vs %r" % (self.shape, other.shape))
            return ShardedDeviceArray([fn(a, b.to(a.device)) for a, b in zip(self.shards, other.shards)])
        if other is None:
            return ShardedDeviceArray([fn(a) for a in self.shards])
        if isinstance(other, (int, float, np.integer, np.floating)):
            return ShardedDeviceArray([fn(a, other) for a in self.shards])
        return NotImplemented

    def __mul__(self, o):
        return self._map(lambda a, b: a * b, o)

    __rmul__ = __mul__

    def __add__(self, o):
        return self._map(lambda a, b: a + b, o)

    __radd__ = __add__

    def __sub__(self, o):
        return self._map(lambda a, b: a - b, o)

    def __rsub__(self, o):
        return self._map(lambda a, b: b - a, o)

    def __truediv__(self, o):
        return self._map(lambda a, b: a / b, o)

    def __neg__(self):
        return self._map(lambda a: -a)

    def abs(self):
        return self._map(lambda a: a.abs())

    def max(self):
        import torch
        return torch.stack([s.max().cpu() for s in self.shards]).max()

    def min(self):
        import torch
        return torch.stack([s.min().cpu() for s in self.shards]).min()


def shard_input(u, devices):
    """The `u` argument of a psampler call, `(n_dev, B, ...)` as a host array, a tensor on any device, a ShardedDeviceArray or a list
    of per-device arrays -> one float32 tensor per device, each resident on its device (peer or host copies, queued, not waited for)."""
    import torch
    n_dev = len(devices)
    if isinstance(u, ShardedDeviceArray):
        parts = u.shards
    elif isinstance(u, (list, tuple)):
        parts = list(u)
    else:
        if not hasattr(u, 'is_cuda'):
            u = torch.as_tensor(np.asarray(u), dtype=torch.float32)
        parts = [u[d] for d in range(u.shape[0])]
    if len(parts) != n_dev:
        raise ValueError("the leading (device) axis of u is %d but this process drives %d device(s) %r: pmap needs one slice per "
                         "local device (cld_jax/sampling.py:236); set GDDIM_LOCAL_DEVICES to change the device list" %
                         (len(parts), n_dev, list(devices)))
    out = []
    for p, d in zip(parts, devices):
        if not hasattr(p, 'is_cuda'):
            p = torch.as_tensor(np.asarray(p), dtype=torch.float32)
        # (without a GPU the slices stay on the host: the engine behind the sampler refuses to run there, so only the axis plumbing
        # of a stand-in sampler ever sees them -- tests/test_devices_cpu.py)
        out.append(p.to(device=torch.device('cuda', d), dtype=torch.float32, non_blocking=True) if torch.cuda.is_available()
                   else p.to(torch.float32))
    return out


def pmap_sampler(sampler, prng, pstate, batch_size, u, sde, data_shape, n_out, **kw):
    """jax.pmap(sampler, static_broadcasted_argnums=2)(prng, pstate, batch_size, u) (cld_jax/sampling.py:236): `sampler(rng, state,
    batch_size, u, slot=d, **kw)` once per local device, each on its own engine (`slot` picks it).  Every call only QUEUES work --
    the staging copy, one hipGraphLaunch, the copy-out -- so device d + 1 is launched while device d computes.  Returns `n_out`
    ShardedDeviceArrays followed by the first device's remaining outputs (`nfes[0]`)."""
    devices = local_devices()
    n_dev = len(devices)
    seeds = device_seeds(prng, n_dev)
    if u is None:
        # the reference draws ONE (n_dev, B, ...) array from the unreplicated key; jax's threefry stream cannot be reproduced anyway,
        # so every device draws its own slice on its own generator (no cross-device copy)
        import torch
        parts = [sde.prior_sampling(seeds[d], (batch_size,) + tuple(data_shape),
                                    device=torch.device('cuda', devices[d]) if torch.cuda.is_available() else torch.device('cpu'))
                 for d in range(n_dev)]
    else:
        parts = shard_input(u, devices)
    if n_dev > 1 and os.environ.get("GDDIM_DEVICE_THREADS", "1").strip() not in ("0", "off", "no", "false"):
        outs = _issue_threaded(sampler, seeds, pstate, batch_size, parts, devices, kw)
    else:
        outs = [sampler(seeds[d], pstate, batch_size, parts[d], slot=d, **kw) for d in range(n_dev)]
    sharded = tuple(ShardedDeviceArray([o[i] for o in outs]) for i in range(n_out))
    return sharded + tuple(outs[0][n_out:])


_ISSUE_POOLS = {}


def _issue_threaded(sampler, seeds, pstate, batch_size, parts, devices, kw):
    """One issuing thread per device.  A sampler call is queued work only, but queueing is not free: hipGraphLaunch of the whole-call
    graph (21 350 kernel nodes at NFE 50) takes 185 ms of host time at B = 256 (358 ms at B = 512; tools/devices_overhead.py), and a
    second launch from the SAME thread waited for the first graph on the one-GPU box -- issued back to back from one thread, device d
    would start d x 185 ms late at best.  The native calls release the GIL (ctypes), every C entry point sets its thread's HIP device
    (DeviceScope), captures are thread-local (hipStreamCaptureModeThreadLocal), engines share nothing.  Each thread works on the stream
    the CALLER had current on its device, so ordering against the caller's earlier and later work is what the serial form gives.
    GDDIM_DEVICE_THREADS=0: the serial form (bit-identical results: tests/test_gpu_devices.py)."""
    import concurrent.futures
    import torch
    n_dev = len(devices)
    pool = _ISSUE_POOLS.get(n_dev)
    if pool is None:
        pool = _ISSUE_POOLS[n_dev] = concurrent.futures.ThreadPoolExecutor(max_workers=n_dev, thread_name_prefix="gddim-dev")
    on_gpu = torch.cuda.is_available()
    streams = [torch.cuda.current_stream(devices[d]) if on_gpu else None for d in range(n_dev)]

    def one(d):
        if not on_gpu:
            return sampler(seeds[d], pstate, batch_size, parts[d], slot=d, **kw)
        with torch.cuda.device(devices[d]), torch.cuda.stream(streams[d]):
            return sampler(seeds[d], pstate, batch_size, parts[d], slot=d, **kw)

    futs = [pool.submit(one, d) for d in range(n_dev)]
    outs, first_error = [], None
    for f in futs:  # wait for every device before raising: no thread keeps issuing into a call that already failed
        try:
            outs.append(f.result())
        except BaseException as exc:  # noqa: BLE001 -- re-raised below
            outs.append(None)
            first_error = first_error or exc
    if first_error is not None:
        raise first_error
    return outs
