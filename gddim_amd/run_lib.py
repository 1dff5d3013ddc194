"""Sampling rounds and the FID report: the evaluation entry points of the reference's run driver.

  sample_data(config, ckpt_file, result_folder, is_continue=False)   cld_jax/run_lib.py:674-731 (blur_jax/run_lib.py:300-345)
  sampling_from_fn(config, state, sampling_fn, result_folder)        cld_jax/run_lib.py:646-671
  check_fid(config, result_folder, ...)                              cld_jax/run_lib.py:735-791

Same files on disk: `samples_<r>.npz` with keys samples (uint8, clip(x*255, 0, 255) truncated), nfe_cnt, samples_v, samples_x;
`statistics_<r>.npz` with pool_3, logits, nfe_cnt; `report.npz` with IS, fid, nfe.  The reference shards one round over
`jax.local_device_count()` devices with pmap from ONE process (cld_jax/run_lib.py:711-725).  Two ways to the same files here:
  * one process, no launcher (the reference's own form): the sampling function drives one engine per local GPU
    (gddim_amd.devices), every device samples `eval.batch_size // n_dev` images, the shards go device -> pinned host memory;
  * one process per GPU under torchrun (WORLD_SIZE > 1): every rank samples its share and the shards meet in one all-gather
    (RCCL) before rank 0 writes the file.
Either way the file is written on writer threads behind pinned staging buffers, so that round r + 1 samples while round r is
compressed (`_RoundWriter`).  Training, logging and the rest of run_lib are out of scope."""
import os
import sys

import numpy as np

from . import checkpoints, datasets, devices, distributed, evaluation
from ._abi import GddimError


def _modules(config):
    """The CLD and blur packages expose the same factory API; the SDE family is read off the config like main.py does."""
    if 'sigma_blur_max' in config.model:  # blur_jax/configs/default_cifar10_config.py; the CLD configs carry m_inv instead
        from .blur import sampling, sde_lib
    else:
        from .cld import sampling, sde_lib
    return sampling, sde_lib


def sample_data(config, ckpt_file, result_folder, is_continue=False, state=None, stats=None):
    """Generate eval.num_samples // eval.batch_size + 1 rounds of eval.batch_size images.  `state` overrides the checkpoint
    (synthetic weights); otherwise a missing `ckpt_file` raises RuntimeError like the reference.  `stats` (a dict) receives the
    wall time, per-round device times and image count of the call (tools/run_50k.py)."""
    import torch
    from .models import utils as mutils
    rank, world, local_rank = distributed.init_from_env()
    os.makedirs(result_folder, exist_ok=True)
    inverse_scaler = datasets.get_data_inverse_scaler(config)
    score_model = mutils.get_model(config.model.name)(config=config)
    sampling, sde_lib = _modules(config)
    sde = sde_lib.from_config(config)
    sampling_fn = sampling.get_sampling_fn(config, sde, score_model, None, inverse_scaler)
    num_sampling_rounds = config.eval.num_samples // config.eval.batch_size + 1
    if state is None:
        if not os.path.exists(ckpt_file):
            raise RuntimeError("%s not exist" % ckpt_file)
        state = checkpoints.restore_checkpoint(ckpt_file, mutils.State(lr=config.optim.lr, ema_rate=config.model.ema_rate))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    try:
        return _rounds(config, state, sampling_fn, result_folder, is_continue, rank, world, stats)
    finally:
        score_model.close()  # this call built the engines (arena + weight copies in HBM); hand them back


def sampling_from_fn(config, state, sampling_fn, result_folder):
    """cld_jax/run_lib.py:646-671: the same rounds with a sampling function the caller built (`sampling.get_sampling_fn(...)` or
    `get_deis_sampler(..., is_p=True)`) and a `State` already in hand; every round is (re)written."""
    import torch
    rank, world, local_rank = distributed.init_from_env()
    os.makedirs(result_folder, exist_ok=True)
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return _rounds(config, state, sampling_fn, result_folder, False, rank, world)


class _RoundWriter:
    """Takes the host side of a round -- device-to-host copies, zlib, the file -- off the sampler's critical path (rank 0).

    The reference's loop is strictly serial: sample, copy, `np.savez_compressed` (25 MB of incompressible float32 noise per 1 024
    images: about a second of zlib), next round (cld_jax/run_lib.py:714-731).  Here a finished round is copied into one of
    `writers + 1` pinned staging slots on a side stream while the next round's sampler call is already queued, and `writers`
    threads (zlib releases the GIL) wait for their slot's copy event, write `samples_<r>.npz.tmp.npz` and rename it -- the same
    bytes, keys and atomic-rename resume semantics as the serial form.  A slot is handed back when its file is on disk, so at most
    `writers + 1` rounds are ever in flight (back-pressure instead of unbounded host memory).  writers = 0 is the serial form."""

    def __init__(self, writers):
        import queue
        import threading
        self.writers = int(writers)
        self.written = {}
        self.error = None
        self._stream = None
        if self.writers <= 0:
            return
        self._free = queue.Queue()
        self._jobs = queue.Queue()
        self._slots = [dict() for _ in range(self.writers + 1)]
        for i in range(len(self._slots)):
            self._free.put(i)
        self._threads = [threading.Thread(target=self._work, name="gddim-round-writer-%d" % i, daemon=True) for i in range(self.writers)]
        for t in self._threads:
            t.start()

    @staticmethod
    def _save(f_sample, keys):
        tmp = f_sample + ".tmp.npz"
        np.savez_compressed(tmp, **keys)
        os.replace(tmp, f_sample)

    def _work(self):
        while True:
            job = self._jobs.get()
            if job is None:
                return
            r, f_sample, slot, events, views, extra = job
            try:
                for event in events:
                    event.synchronize()
                keys = {k: t.numpy().reshape(shape) for k, (t, shape) in views.items()}
                keys.update(extra)
                self._save(f_sample, keys)
                self.written[r] = f_sample
            except BaseException as e:  # surfaced by the next submit() / close() on the main thread
                self.error = e
            finally:
                self._free.put(slot)

    def submit(self, r, f_sample, tensors, extra):
        """tensors: {key: (tensor or list of equally shaped shards -- one per device --, shape on disk)}; extra: {key: small host value}.
        Returns once the copies are QUEUED (every shard on a side stream of its own device)."""
        import torch
        if self.error is not None:
            raise self.error
        tensors = {k: (list(t) if isinstance(t, (list, tuple)) else [t], shape) for k, (t, shape) in tensors.items()}
        if self.writers <= 0:
            keys = {k: np.stack([p.cpu().numpy() for p in parts]).reshape(shape) for k, (parts, shape) in tensors.items()}
            keys.update(extra)
            self._save(f_sample, keys)
            self.written[r] = f_sample
            return
        slot = self._free.get()  # blocks while writers + 1 rounds are in flight
        stage = self._slots[slot]
        gpus = sorted({p.device.index for parts, _ in tensors.values() for p in parts if p.is_cuda})
        if self._stream is None:
            self._stream = {}
        for d in gpus:
            if d not in self._stream:
                self._stream[d] = torch.cuda.Stream(device=d)
            self._stream[d].wait_stream(torch.cuda.current_stream(d))
        views = {}
        for k, (parts, shape) in tensors.items():
            full = (len(parts),) + tuple(parts[0].shape)
            buf = stage.get(k)
            if buf is None or tuple(buf.shape) != full or buf.dtype != parts[0].dtype:
                buf = stage[k] = torch.empty(full, dtype=parts[0].dtype, device="cpu", pin_memory=bool(gpus))
            for i, t in enumerate(parts):
                if t.is_cuda:
                    side = self._stream[t.device.index]
                    with torch.cuda.stream(side):
                        buf[i].copy_(t, non_blocking=True)
                    t.record_stream(side)  # the caching allocator must not hand this block out before the copy has run
                else:
                    buf[i].copy_(t)
            views[k] = (buf, shape)
        events = []
        for d in gpus:
            ev = torch.cuda.Event()
            ev.record(self._stream[d])
            events.append(ev)
        self._jobs.put((r, f_sample, slot, events, views, extra))

    def close(self):
        if self.writers > 0:
            for _ in self._threads:
                self._jobs.put(None)
            for t in self._threads:
                t.join()
        if self.error is not None:
            raise self.error
        return [self.written[r] for r in sorted(self.written)]


def _writer_threads(config):
    """`config.eval.writer_threads` if the config carries it, else GDDIM_WRITER_THREADS, else 2 (0 = the reference's serial loop)."""
    ev = config.eval
    if 'writer_threads' in ev:
        return int(ev.writer_threads)
    return int(os.environ.get("GDDIM_WRITER_THREADS", "2"))


def _shards(a):
    """The per-device slices of one output of a psampler call: a ShardedDeviceArray, or any array with a leading device axis."""
    if isinstance(a, devices.ShardedDeviceArray):
        return list(a.shards)
    return [a[d] for d in range(a.shape[0])]


def _rounds(config, state, sampling_fn, result_folder, is_continue, rank, world, stats=None):
    import time
    import torch
    num_sampling_rounds = config.eval.num_samples // config.eval.batch_size + 1
    if config.eval.batch_size < world:
        raise GddimError("eval.batch_size %d is smaller than the number of ranks %d" % (config.eval.batch_size, world))
    if world > 1:
        # one process per GPU.  The reference gives every device eval.batch_size // n_dev images and silently drops the remainder
        # (cld_jax/run_lib.py:721); here the remainder goes to the first ranks so that a round always holds eval.batch_size images
        n_dev = 1
        _, per_call = distributed.shard_batch(config.eval.batch_size, world, rank)
        equal = config.eval.batch_size % world == 0
    else:
        # the reference's own form: this process drives every local device, each takes eval.batch_size // n_dev images (:721)
        n_dev = devices.local_device_count()
        per_call = config.eval.batch_size // n_dev
        equal = True
        if per_call < 1:
            raise GddimError("eval.batch_size %d is smaller than the number of local devices %d" % (config.eval.batch_size, n_dev))
    writer = _RoundWriter(_writer_threads(config) if rank == 0 else 0)
    gpu = torch.cuda.is_available()
    marks = []
    img_shape = (config.data.image_size, config.data.image_size, config.data.num_channels)
    t_start = time.perf_counter()
    try:
        for r in range(num_sampling_rounds):
            # a writer that failed stops the run at the next round on EVERY rank (the others would wait in the round's collective)
            if distributed.any_rank_failed(writer.error is not None):
                raise writer.error if writer.error is not None else GddimError("another rank's round writer failed")
            f_sample = os.path.join(result_folder, "samples_%d.npz" % r)
            if os.path.exists(f_sample) and is_continue:
                continue
            # one key per device of this call (cld_jax/run_lib.py:715-720: jax.random.split(rng, n_dev + 1)[1:])
            keys = [distributed.rank_seed(config.seed + 1, r, rank * n_dev + d) for d in range(n_dev)]
            out = sampling_fn(keys, state, per_call)
            nfe_cnt = out[-1]
            xs = _shards(out[0])
            vs = _shards(out[1]) if len(out) == 3 else None
            if world > 1:
                x = xs[0]
                u8 = distributed.all_gather_samples(distributed.to_uint8(x), equal_shards=equal)
                # the float copies the reference also stores (samples_x, samples_v): one more gather, x and v stacked
                fl = x[..., None] if vs is None else torch.stack([x, vs[0].reshape(x.shape)], dim=-1)
                fl = distributed.all_gather_samples(fl, equal_shards=equal)
                # reference layout of the float arrays: (n_dev, B / n_dev, H, W, C); uneven shards (not possible there) are written as one row
                lead = (world, config.eval.batch_size // world) if equal else (1, config.eval.batch_size)
                if rank == 0:
                    xf = fl[..., 0].contiguous()
                    tensors = dict(samples=(u8, (-1,) + img_shape), samples_x=(xf, lead + tuple(xf.shape[1:])))
                    if vs is not None:
                        vf = fl[..., 1].contiguous()
                        tensors['samples_v'] = (vf, lead + tuple(vf.shape[1:]))
                    writer.submit(r, f_sample, tensors, dict(nfe_cnt=nfe_cnt))
            else:
                # no collective: every shard goes from its own device to the pinned staging buffer (the reference's host concatenation,
                # cld_jax/run_lib.py:722-725); the uint8 conversion runs where the shard lives
                lead = (n_dev, per_call)
                tensors = dict(samples=([distributed.to_uint8(x) for x in xs], (-1,) + img_shape),
                               samples_x=([x.contiguous() for x in xs], lead + tuple(xs[0].shape[1:])))
                if vs is not None:
                    tensors['samples_v'] = ([v.reshape(x.shape).contiguous() for v, x in zip(vs, xs)], lead + tuple(xs[0].shape[1:]))
                writer.submit(r, f_sample, tensors, dict(nfe_cnt=nfe_cnt))
            if stats is not None:
                if gpu:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    marks.append((r, ev, time.perf_counter()))
                else:
                    marks.append((r, None, time.perf_counter()))
    except BaseException:
        # the loop's own failure is what the caller must see; a writer error on top of it is reported, not raised over it
        try:
            writer.close()
        except BaseException as werr:  # noqa: B902
            sys.stderr.write("gddim_amd.run_lib: the round writer failed as well: %r\n" % (werr,))
        raise
    written = writer.close() if rank == 0 else []
    if stats is not None:
        if gpu:
            for d in devices.local_devices() if world == 1 else [torch.cuda.current_device()]:
                torch.cuda.synchronize(d)
        stats["wall_s"] = time.perf_counter() - t_start
        stats["rounds"] = [m[0] for m in marks]
        stats["host_submit_s"] = [m[2] - t_start for m in marks]
        if gpu and len(marks) > 1:  # GPU time between the ends of consecutive rounds' device work
            stats["gpu_round_ms"] = [marks[i][1].elapsed_time(marks[i + 1][1]) for i in range(len(marks) - 1)]
        stats["images"] = len(marks) * n_dev * per_call if world == 1 else len(marks) * config.eval.batch_size
        stats["writer_threads"] = writer.writers
        stats["local_devices"] = n_dev
    return written


def check_fid(config, result_folder, inception_fn=None, stats_root=None):
    """FID / IS over the rounds of `result_folder`.  inception_fn(uint8 images (N,H,W,C)) -> {'pool_3': (N,2048), 'logits':
    (N,1008)} stands for evaluation.run_inception_distributed; without it the `statistics_<r>.npz` files must already exist."""
    inceptionv3 = config.data.image_size >= 256
    all_logits, all_pools, all_nfe_cnt = [], [], []
    num_sampling_rounds = config.eval.num_samples // config.eval.batch_size + 1
    for r in range(num_sampling_rounds):
        f_stat = os.path.join(result_folder, "statistics_%d.npz" % r)
        if inception_fn is not None:
            info = np.load(os.path.join(result_folder, "samples_%d.npz" % r))
            latents = inception_fn(info["samples"])
            cur_nfe = info["nfe_cnt"]
            np.savez_compressed(f_stat, pool_3=latents["pool_3"], logits=latents.get("logits", np.zeros((0,))), nfe_cnt=cur_nfe)
        elif os.path.exists(f_stat):
            latents = np.load(f_stat)
            cur_nfe = latents["nfe_cnt"]
        else:
            raise GddimError("%s is missing and no inception_fn was given: the TF-Hub Inception network "
                             "(tfgan/eval/inception/1) is an external asset that is not part of this package" % f_stat)
        if not inceptionv3:
            all_logits.append(latents["logits"])
        all_pools.append(latents["pool_3"])
        all_nfe_cnt.append(cur_nfe)
    all_pools = np.concatenate(all_pools, axis=0)[:config.eval.num_samples]
    data_pools = evaluation.load_dataset_stats(config, stats_root)["pool_3"]
    if not inceptionv3:
        inception_score = evaluation.classifier_score_from_logits(np.concatenate(all_logits, axis=0)[:config.eval.num_samples])
    else:
        inception_score = -1
    fid = evaluation.frechet_classifier_distance_from_activations(data_pools, all_pools)
    avg_nfe_cnt = float(np.mean(all_nfe_cnt))
    np.savez_compressed(os.path.join(result_folder, "report.npz"), IS=inception_score, fid=fid, nfe=avg_nfe_cnt)
    return {"IS": inception_score, "fid": fid, "nfe": avg_nfe_cnt}
