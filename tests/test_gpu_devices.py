"""The reference's pmap contract on real engines (cld_jax/sampling.py:232-237, blur_jax/sampling.py:82-87): ONE process, n_dev
engines.  The test box has one MI355X, so `GDDIM_LOCAL_DEVICES=0,0` puts two engines on it -- two arenas, two weight copies, two
streams, two whole-call hipGraphs in flight at once -- and the sharded result must equal two single-device calls bit for bit.  Also
here: the hoisted time-embedding table and the blur whole-call graph, both against the forms they replace, bit for bit."""
import io

import numpy as np
import pytest
import torch

from conftest import tiny_config
from oracle import ncsnpp as onet

pytestmark = pytest.mark.gpu


def _cld(cfg, seed=3):
    from gddim_amd.cld import sde_lib
    from gddim_amd.models import utils as mutils
    params = onet.random_params(cfg, 6, seed=seed)
    return params, sde_lib.from_config(cfg), mutils.get_model(cfg.model.name)(config=cfg), mutils.State(params_ema=params, model_state={})


def _noise(n_dev, B, size, seed=0):
    rng = np.random.default_rng(seed)
    shape = (n_dev, B, size, size, 3)
    return np.stack([rng.standard_normal(shape), rng.standard_normal(shape) / 2], axis=-1).astype(np.float32)


@pytest.mark.parametrize("full", [False, True], ids=["tiny", "accr-B32"])
def test_cld_psampler_two_engines_equal_two_single_calls(native_lib, monkeypatch, full):
    """(a) of VERDICT r5 item 1.  `full`: the benchmarked accr network at B = 32 per engine -- the production kernels of two engines
    interleave on one GPU."""
    from gddim_amd import devices
    from gddim_amd.cld import sampling
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.datasets import get_data_inverse_scaler
    cfg = cld_accr_dcifar10.get_config() if full else tiny_config("cld", nf=16, num_res_blocks=1, image_size=16, ch_mult=(1, 2, 2), attn=(8,))
    size, B, nfe = cfg.data.image_size, (32 if full else 4), (3 if full else 6)
    cfg.sampling.method, cfg.sampling.nfe, cfg.sampling.deis_order, cfg.sampling.noise_removal = 'deis', nfe, 2, True
    params, sde, model, state = _cld(cfg)
    u = _noise(2, B, size)
    monkeypatch.setenv("GDDIM_LOCAL_DEVICES", "0,0")
    assert devices.local_device_count() == 2
    pfn = sampling.get_sampling_fn(cfg, sde, model, None, get_data_inverse_scaler(cfg))
    xs, vs, n = pfn([11, 12], state, B, u)  # a host array with the device axis, as the reference's callers pass it
    assert isinstance(xs, devices.ShardedDeviceArray) and xs.shape == (2, B, size, size, 3) == vs.shape and n == nfe
    assert len(model._engines) == 2 and {k[2] for k in model._engines} == {0, 1}  # one engine per slot, both on device 0
    # again, replayed from the two captured graphs, input already on the device
    xs2, vs2, _ = pfn([11, 12], state, B, torch.as_tensor(u, device="cuda:0"))
    assert all(torch.equal(a, b) for a, b in zip(xs.shards + vs.shards, xs2.shards + vs2.shards))
    # ... and against single-device calls of the non-pmapped sampler on a model of its own
    monkeypatch.setenv("GDDIM_LOCAL_DEVICES", "0")
    _, _, model1, state1 = _cld(cfg)
    fn = sampling.get_deis_sampler(sde, model1, (size, size, 3), nfe, get_data_inverse_scaler(cfg), deis_order=2, ts_order=cfg.sampling.ts_order,
                                   denoising=True, is_p=False)
    for d in range(2):
        x, v, _ = fn(0, state1, B, torch.as_tensor(u[d], device="cuda:0"))
        assert torch.isfinite(x).all() and torch.equal(xs[d], x) and torch.equal(vs[d], v), "device slice %d differs from the single-device call" % d
    assert not torch.equal(xs[0], xs[1])
    # one device, the same factory: the device axis has length 1 and u with two slices is refused the way pmap refuses it
    p1 = sampling.get_sampling_fn(cfg, sde, model1, None, get_data_inverse_scaler(cfg))
    x1, _, _ = p1([0], state1, B, u[:1])
    assert x1.shape == (1, B, size, size, 3) and torch.equal(x1[0], xs[0])
    with pytest.raises(ValueError, match="device"):
        p1([0], state1, B, u)
    # the reference's caller, verbatim (cld_jax/run_lib.py:723-730)
    samples_org_x, samples_v, nfe_cnt = xs, vs, n
    samples_x = np.clip(samples_org_x * 255., 0, 255).astype(np.uint8)
    samples_x = samples_x.reshape((-1, cfg.data.image_size, cfg.data.image_size, cfg.data.num_channels))
    io_buffer = io.BytesIO()
    np.savez_compressed(io_buffer, samples=samples_x, nfe_cnt=nfe_cnt, samples_v=samples_v, samples_x=samples_org_x)
    io_buffer.seek(0)
    info = np.load(io_buffer)
    assert info["samples"].shape == (2 * B, size, size, 3) and info["samples_x"].shape == (2, B, size, size, 3)
    assert np.array_equal(info["samples_x"][1], xs[1].cpu().numpy())
    model.close()
    model1.close()


def test_cld_psampler_draws_a_prior_per_device(native_lib, monkeypatch):
    """u=None: every device draws its own slice from its own key; the stochastic samplers' per-call seeds differ per device too."""
    from gddim_amd.cld import sampling
    from gddim_amd.datasets import get_data_inverse_scaler
    cfg = tiny_config("cld", nf=16, num_res_blocks=1, image_size=16, ch_mult=(1, 2, 2), attn=(8,))
    cfg.sampling.method, cfg.sampling.nfe, cfg.sampling.noise_removal = 'em', 4, True
    cfg.sampling.lambda_coef = 1.0
    params, sde, model, state = _cld(cfg)
    monkeypatch.setenv("GDDIM_LOCAL_DEVICES", "0,0")
    pfn = sampling.get_sampling_fn(cfg, sde, model, None, get_data_inverse_scaler(cfg))
    xs, vs, n = pfn([21, 22], state, 2)
    assert xs.shape == (2, 2, 16, 16, 3) and torch.isfinite(xs[0]).all() and not torch.equal(xs[0], xs[1])
    xs2, _, _ = pfn([21, 22], state, 2)
    assert torch.equal(xs2[0], xs[0]) and torch.equal(xs2[1], xs[1])  # the keys decide
    u = _noise(2, 2, 16, seed=5)
    u[1] = u[0]  # the same initial state on both devices: only the per-device key separates the two stochastic trajectories
    xa, _, _ = pfn([21, 22], state, 2, u)
    assert not torch.equal(xa[0], xa[1])
    xb, _, _ = pfn([21, 21], state, 2, u)
    assert torch.equal(xb[0], xb[1]) and torch.equal(xb[0], xa[0])
    model.close()


def _blur(cfg, seed=5):
    from gddim_amd.blur import sde_lib
    from gddim_amd.models import utils as mutils
    params = onet.random_params(cfg, 3, seed=seed)
    return params, sde_lib.from_config(cfg), mutils.get_model(cfg.model.name)(config=cfg), mutils.State(params_ema=params, model_state={})


def test_blur_psampler_two_engines_and_whole_call_graph(native_lib, monkeypatch):
    """blur_jax/sampling.py:82-87 with n_dev = 2, and the three launch forms of the blur loop -- eager, one graph per step, ONE graph per
    call (the default since round 6) -- bit for bit, order0 and the multistep extension."""
    from gddim_amd import devices
    from gddim_amd.blur import sampling
    from gddim_amd.datasets import get_data_inverse_scaler
    cfg = tiny_config("blur", nf=16, num_res_blocks=1, image_size=32, ch_mult=(1, 2, 2), attn=(16,))
    cfg.sampling.nfe = 5
    params, sde, model, state = _blur(cfg)
    inv = get_data_inverse_scaler(cfg)
    y = np.random.default_rng(2).standard_normal((2, 3, 32, 32, 3)).astype(np.float32)
    assert sampling.DEFAULT_GRAPH == 'loop'
    for order in (None, 2):
        fn = (sampling.get_order0_sampler(sde, model, (32, 32, 3), ts_order=2, nfe=5, inverse_scaler=inv) if order is None else
              sampling.get_deis_sampler(sde, model, (32, 32, 3), ts_order=2, nfe=5, deis_order=order, inverse_scaler=inv))
        yt = torch.as_tensor(y[0], device="cuda:0")
        outs = {mode: fn(0, state, 3, yt, use_graph=mode)[0] for mode in (False, True, 'loop')}
        assert torch.isfinite(outs['loop']).all() and torch.equal(outs[False], outs[True]) and torch.equal(outs[False], outs['loop'])
        assert torch.equal(fn(0, state, 3, yt)[0], outs['loop']) and torch.equal(fn(0, state, 3, yt, use_graph='loop')[0], outs['loop'])  # default; replay
        assert torch.equal(fn(0, state, 2, yt[:2], use_graph='loop')[0], outs['loop'][:2])  # another batch size: its own graph
    monkeypatch.setenv("GDDIM_LOCAL_DEVICES", "0,0")
    pfn = sampling.get_sampling_fn(cfg, sde, model, None, inv)
    xs, n = pfn([1, 2], state, 3, y)
    assert isinstance(xs, devices.ShardedDeviceArray) and xs.shape == (2, 3, 32, 32, 3) and n == 5
    monkeypatch.setenv("GDDIM_LOCAL_DEVICES", "0")
    one = sampling.get_order0_sampler(sde, model, (32, 32, 3), ts_order=2, nfe=5, inverse_scaler=inv)
    for d in range(2):
        assert torch.equal(xs[d], one(0, state, 3, torch.as_tensor(y[d], device="cuda:0"))[0])
    model.close()


@pytest.mark.parametrize("kind", ["cld-accr", "cld-tiny-positional", "blur"])
def test_time_embedding_table_is_bit_identical_to_the_in_loop_embedding(native_lib, gd_env, kind):
    """SURVEY K4 / VERDICT r5 item 3a: Fourier -> Dense_0 -> Dense_1 -> the concatenated per-block projection run once per knot when the
    schedule is installed; the loop reads row `step`.  Against an engine created under GDDIM_NO_TEMB_HOIST=1 (the embedding inside every
    step): the same bits in all three launch forms, and the embedding kernels are gone from a sampler call's profile."""
    from gddim_amd.datasets import get_data_inverse_scaler
    if kind == "blur":
        from gddim_amd.blur import sampling
        cfg = tiny_config("blur", nf=16, num_res_blocks=1, image_size=32, ch_mult=(1, 2, 2), attn=(16,))
        make = lambda: _blur(cfg)  # noqa: E731
        B, in_ch = 3, 3
        u = torch.as_tensor(np.random.default_rng(3).standard_normal((B, 32, 32, 3)).astype(np.float32), device="cuda:0")
        build = lambda sde, model: sampling.get_order0_sampler(sde, model, (32, 32, 3), ts_order=2, nfe=4, inverse_scaler=get_data_inverse_scaler(cfg))  # noqa: E731
    else:
        from gddim_amd.cld import sampling
        from gddim_amd.configs import cld_accr_dcifar10
        if kind == "cld-accr":
            cfg, B = cld_accr_dcifar10.get_config(), 8
        else:
            cfg, B = tiny_config("cld", nf=16, num_res_blocks=1, image_size=16, ch_mult=(1, 2, 2), attn=(8,)), 4
            cfg.model.embedding_type = 'positional'
        make = lambda: _cld(cfg)  # noqa: E731
        in_ch = 6
        u = torch.as_tensor(_noise(1, B, cfg.data.image_size, seed=4)[0], device="cuda:0")
        build = lambda sde, model: sampling.get_deis_sampler(sde, model, (cfg.data.image_size,) * 2 + (3,), 4, get_data_inverse_scaler(cfg),  # noqa: E731
                                                             deis_order=2, denoising=True)
    results = {}
    for hoist in (True, False):
        gd_env(GDDIM_NO_TEMB_HOIST=None if hoist else "1")
        _, sde, model, state = make()
        fn = build(sde, model)
        outs = [fn(0, state, B, u, use_graph=mode) for mode in (False, True, 'loop')]
        eng = model._engines[(in_ch, 0, 0)]['eng']
        assert eng.plan()["options"]["GDDIM_NO_TEMB_HOIST"] == (0 if hoist else 1)
        eng.profile(True)
        fn(0, state, B, u, use_graph=False)
        names = sorted({p["name"] for p in eng.profile_read()})
        eng.profile(False)
        results[hoist] = (outs, names)
        # a new schedule and new weights rebuild the table: another NFE on the same engine, then the first one again
        if hoist:
            again = fn(0, state, B, u)
            assert all(torch.equal(a, b) for a, b in zip(again[:-1], outs[2][:-1]))
        model.close()
    (oh, nh), (on, nn) = results[True], results[False]
    for a, b in zip(oh, on):
        assert all(torch.isfinite(t).all() and torch.equal(t, w) for t, w in zip(a[:-1], b[:-1]))
    assert all(torch.equal(t, w) for o in oh[1:] for t, w in zip(o[:-1], oh[0][:-1]))  # eager = per-step graph = whole-call graph
    assert "fourier" in nn and "fourier" not in nh, (nh, nn)
    gone = set(nn) - set(nh)
    print("%s: kernels no longer in a sampler call: %s" % (kind, sorted(gone)))
    assert set(nh) <= set(nn)


def test_time_embedding_table_follows_schedule_and_weights(native_lib):
    """The table is rebuilt when the schedule or the parameter tree changes under a live engine (and only then)."""
    from gddim_amd.cld import sampling
    from gddim_amd.datasets import get_data_inverse_scaler
    cfg = tiny_config("cld", nf=16, num_res_blocks=1, image_size=16, ch_mult=(1, 2, 2), attn=(8,))
    params, sde, model, state = _cld(cfg)
    inv = get_data_inverse_scaler(cfg)
    u = torch.as_tensor(_noise(1, 4, 16, seed=6)[0], device="cuda:0")
    f4 = sampling.get_deis_sampler(sde, model, (16, 16, 3), 4, inv, deis_order=1, denoising=True)
    f7 = sampling.get_deis_sampler(sde, model, (16, 16, 3), 7, inv, deis_order=1, denoising=False)
    a4, a7 = f4(0, state, 4, u)[0], f7(0, state, 4, u)[0]
    b4 = f4(0, state, 4, u)[0]  # back to the first schedule on the same engine
    assert torch.equal(a4, b4) and not torch.equal(a4, a7)
    # other weights in the same engine: equal to a fresh model with those weights
    params2 = onet.random_params(cfg, 6, seed=9)
    from gddim_amd.models import utils as mutils
    state2 = mutils.State(params_ema=params2, model_state={})
    c4 = f4(0, state2, 4, u)[0]
    fresh = mutils.get_model(cfg.model.name)(config=cfg)
    g4 = sampling.get_deis_sampler(sde, fresh, (16, 16, 3), 4, inv, deis_order=1, denoising=True)(0, state2, 4, u)[0]
    assert torch.equal(c4, g4) and not torch.equal(c4, a4)
    # per-sample labels (gd_cld_eps) never use the table
    t = torch.tensor([0.1, 0.5, 0.9, 0.3], device="cuda:0")
    e1 = mutils.get_eps_fn(sde, model, params2, {})(u, t)
    e2 = mutils.get_eps_fn(sde, fresh, params2, {})(u, t)
    assert torch.equal(e1, e2)
    model.close()
    fresh.close()


def test_psampler_threaded_issue_equals_serial_issue(native_lib, monkeypatch):
    """One issuing thread per device (the default for n_dev > 1: hipGraphLaunch of the whole-call graph costs the host 185 ms at B = 256,
    and issued from one thread device d would start d launches late) against the calls back to back from the calling thread
    (GDDIM_DEVICE_THREADS=0): bit-identical shards, first call (engine creation, weight upload, graph capture in the threads) and replay,
    CLD and blur, the input produced on a non-default stream of the caller."""
    from gddim_amd import devices
    from gddim_amd.cld import sampling
    from gddim_amd.datasets import get_data_inverse_scaler
    cfg = tiny_config("cld", nf=16, num_res_blocks=1, image_size=16, ch_mult=(1, 2, 2), attn=(8,))
    cfg.sampling.method, cfg.sampling.nfe, cfg.sampling.deis_order, cfg.sampling.noise_removal = 'deis', 6, 2, True
    monkeypatch.setenv("GDDIM_LOCAL_DEVICES", "0,0,0")
    res = {}
    side = torch.cuda.Stream(device=0)
    for threads in ("1", "0"):
        monkeypatch.setenv("GDDIM_DEVICE_THREADS", threads)
        params, sde, model, state = _cld(cfg)
        pfn = sampling.get_sampling_fn(cfg, sde, model, None, get_data_inverse_scaler(cfg))
        with torch.cuda.stream(side):  # the caller's stream: the threads must order their work on it
            u = torch.as_tensor(_noise(3, 4, 16), device="cuda:0") * 1.0
            first = pfn([5, 6, 7], state, 4, u)
            again = pfn([5, 6, 7], state, 4, u)
            got = [t.clone() for t in first[0].shards + first[1].shards + again[0].shards + again[1].shards]
        side.synchronize()
        assert len(model._engines) == 3 and devices.local_device_count() == 3
        res[threads] = got
        model.close()
    assert all(torch.isfinite(t).all() for t in res["1"])
    assert all(torch.equal(a, b) for a, b in zip(res["1"], res["0"]))
    assert all(torch.equal(a, b) for a, b in zip(res["1"][:6], res["1"][6:]))  # replay = first call
    assert not torch.equal(res["1"][0], res["1"][1])


def test_psampler_threaded_issue_surfaces_a_device_error(native_lib, monkeypatch):
    """An error in one device's call reaches the caller after every thread has finished (no thread keeps issuing into a failed call)."""
    from gddim_amd import devices
    monkeypatch.setenv("GDDIM_LOCAL_DEVICES", "0,0")
    monkeypatch.setenv("GDDIM_DEVICE_THREADS", "1")
    seen = []

    def sampler(seed, state, B, u, slot=0):
        seen.append(slot)
        if slot == 1:
            raise RuntimeError("device 1 failed")
        return (torch.zeros(2, device="cuda:0"), torch.zeros(2, device="cuda:0"), 3)

    with pytest.raises(RuntimeError, match="device 1 failed"):
        devices.pmap_sampler(sampler, [1, 2], None, 2, torch.zeros((2, 2, 4, 4, 3, 2), device="cuda:0"), None, (4, 4, 3, 2), 2)
    assert sorted(seen) == [0, 1]
