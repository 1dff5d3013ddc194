"""GPU parity of the NCSN++ forward (gd_unet_forward / gd_cld_eps) against the torch-CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import tiny_config
from oracle import ncsnpp as onet

pytestmark = pytest.mark.gpu


def _forward_pair(cfg, in_ch, B, seed, mixed_t=True):
    from gddim_amd.engine import Engine
    params = onet.random_params(cfg, in_ch, seed=seed)
    eng = Engine(cfg, in_ch, max_batch=B, device=0)
    eng.load_params(params)
    rng = np.random.default_rng(seed + 1)
    H = cfg.data.image_size
    x = rng.standard_normal((B, H, H, in_ch)).astype(np.float32)
    t = rng.uniform(1e-3, 1.0, size=B).astype(np.float32) if mixed_t else np.full(B, 0.37, np.float32)
    labels = (t * 999).astype(np.float32)
    got = eng.unet(torch.as_tensor(x, device="cuda:0"), torch.as_tensor(labels, device="cuda:0")).cpu().numpy()
    with torch.no_grad():
        want64 = onet.ncsnpp_forward(params, torch.as_tensor(x, dtype=torch.float64), torch.as_tensor(labels, dtype=torch.float64), cfg).numpy()
    return eng, params, x, t, got, want64


@pytest.mark.parametrize("kind,in_ch", [("cld", 6), ("blur", 3)])
def test_unet_tiny_matches_oracle(native_lib, kind, in_ch):
    eng, params, x, t, got, want = _forward_pair(tiny_config(kind), in_ch, B=3, seed=1)
    scale = max(1.0, np.abs(want).max())
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() < 1e-4 * scale, np.abs(got - want).max()
    eng.close()


def test_unet_variants_match_oracle(native_lib):
    cfg = tiny_config("cld", nf=32, num_res_blocks=2, image_size=32, ch_mult=(1, 2, 2, 2), attn=(16,), centered=False)
    eng, params, x, t, got, want = _forward_pair(cfg, 6, B=2, seed=4)
    assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max())
    eng.close()


def test_unet_standalone_statistics_fallback(native_lib):
    """8x8 input: the 2x2 level uses the stand-alone GroupNorm statistics kernel, the others the fused epilogue statistics."""
    cfg = tiny_config("cld", nf=32, num_res_blocks=1, image_size=8, ch_mult=(1, 2, 2), attn=(4,))
    eng, params, x, t, got, want = _forward_pair(cfg, 6, B=5, seed=12)
    assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max())
    eng.close()


def test_unet_full_architecture_matches_oracle(native_lib):
    """The benchmarked accr_dcifar10 architecture (107.6 M parameters), per-sample t, against float64 and against
    the float32 oracle: the HIP forward must sit as close to float64 as the float32 oracle does (x4)."""
    from gddim_amd.configs import cld_accr_dcifar10
    cfg = cld_accr_dcifar10.get_config()
    eng, params, x, t, got, want64 = _forward_pair(cfg, 6, B=2, seed=0)
    labels = (t * 999).astype(np.float32)
    with torch.no_grad():
        want32 = onet.ncsnpp_forward(params, torch.as_tensor(x), torch.as_tensor(labels), cfg).numpy()
    scale = max(1.0, np.abs(want64).max())
    e_hip = np.abs(got - want64).max() / scale
    e_f32 = np.abs(want32 - want64).max() / scale
    print("full arch: |hip-f64| %.3e  |f32oracle-f64| %.3e  scale %.3f" % (e_hip, e_f32, scale))
    assert e_hip < max(4 * e_f32, 2e-5), (e_hip, e_f32)
    # cld eps wrapper: pack -> unet -> unpack equals the planar call
    u = np.stack([x[..., :3], x[..., 3:]], axis=-1)  # (B,H,W,3,2)
    eps = eng.cld_eps(torch.as_tensor(u, device="cuda:0"), torch.as_tensor(t, device="cuda:0")).cpu().numpy()
    want_eps = onet.cld_eps_fn(params, cfg, dtype=torch.float64)(u.astype(np.float64), t.astype(np.float64))
    assert np.abs(eps - want_eps).max() < max(4 * e_f32, 2e-5) * scale
    # weight footprint in HBM: the packed GEMM layout (430 MB) + ONE Winograd copy per 3x3 slot -- hybrid fragment order for the 64x64 .. 8x8
    # levels, F(2x2,3x3) for the 4x4 level -- 2.8 GB in round 3, when every slot also carried an F(2,3) and an F(2x2,3x3) copy (<= 2.2 GB)
    wb = eng.weight_bytes()
    print("weights in HBM: %s -> %.2f GB" % (wb, sum(wb.values()) / 1e9))
    assert 4.2e8 < wb["packed"] < 4.5e8 and wb["bf16_split"] == 0
    assert 0 < wb["wino2d"] < 0.3e9 and 0.6e9 < wb["wino42"] < 1.0e9 and sum(wb.values()) < 2.2e9
    eng.close()


def test_unet_batch_independence_and_determinism(native_lib):
    """Every image is an independent unit (GroupNorm per sample, attention per image): running a batch or its
    images one by one gives bit-identical results, and repeated runs are bit-identical."""
    cfg = tiny_config("cld")
    eng, params, x, t, got, want = _forward_pair(cfg, 6, B=4, seed=9)
    labels = torch.as_tensor((t * 999).astype(np.float32), device="cuda:0")
    xt = torch.as_tensor(x, device="cuda:0")
    again = eng.unet(xt, labels).cpu().numpy()
    assert np.array_equal(again, got)
    one = eng.unet(xt[1:2], labels[1:2]).cpu().numpy()
    assert np.abs(one[0] - got[1]).max() < 1e-5 * max(1.0, np.abs(got).max())
    eng.close()


def test_compute_fails_loudly_without_params(native_lib):
    from gddim_amd._abi import GddimError
    from gddim_amd.engine import Engine
    eng = Engine(tiny_config("cld"), 6, max_batch=1, device=0)
    with pytest.raises(GddimError):
        eng.unet(torch.zeros(1, 16, 16, 6, device="cuda:0"), torch.ones(1, device="cuda:0"))
    eng.close()


@pytest.mark.parametrize("nf,nrb,size", [(64, 4, 32), (32, 2, 16)])
def test_unet_ddpmpp_style_matches_oracle(native_lib, nf, nrb, size):
    """configs/ddpmpp_cifar10_config.py / debug_ / simple_: positional embedding, fir=False, progressive_input='none',
    num_res_blocks=4 (first case: the reference's debug config at full depth, nf=64)."""
    cfg = tiny_config("cld", nf=nf, num_res_blocks=nrb, image_size=size, ch_mult=(1, 2, 2, 2) if size == 32 else (1, 2, 2),
                      attn=(16,) if size == 32 else (8,), progressive_input='none')
    cfg.model.fir = False
    cfg.model.embedding_type = 'positional'
    eng, params, x, t, got, want = _forward_pair(cfg, 6, B=3, seed=21)
    assert np.isfinite(got).all() and np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max()), np.abs(got - want).max()
    eng.set_precision("bf16x3")
    got3 = eng.unet(torch.as_tensor(x, device="cuda:0"), torch.as_tensor((t * 999).astype(np.float32), device="cuda:0")).cpu().numpy()
    assert np.abs(got3 - want).max() < 1e-4 * max(1.0, np.abs(want).max())
    eng.close()


def test_unet_64x64_winograd_levels_match_oracle(native_lib):
    """BASELINE config 5 geometry (64x64 input, nf = 128): at batch 16 the upper levels run the two-dimensional Winograd kernel (asserted
    from the engine profile), the lower ones the direct kernel.  Images are independent units: the first image of the batch against the
    float64 oracle of that image (bar: 4x the float32 oracle's own error), and against the same image evaluated alone (direct kernels)."""
    from gddim_amd.engine import Engine
    cfg = tiny_config("cld", nf=128, num_res_blocks=1, image_size=64, ch_mult=(1, 2, 2, 2), attn=(16,))
    params = onet.random_params(cfg, 6, seed=3)
    rng = np.random.default_rng(5)
    B = 16
    x = rng.standard_normal((B, 64, 64, 6)).astype(np.float32)
    labels = (rng.uniform(1e-3, 1.0, size=B) * 999).astype(np.float32)
    with torch.no_grad():
        want64 = onet.ncsnpp_forward(params, torch.as_tensor(x[:1], dtype=torch.float64), torch.as_tensor(labels[:1], dtype=torch.float64), cfg).numpy()
        want32 = onet.ncsnpp_forward(params, torch.as_tensor(x[:1]), torch.as_tensor(labels[:1]), cfg).numpy()
    scale = max(1.0, np.abs(want64).max())
    e_f32 = np.abs(want32 - want64).max() / scale
    eng = Engine(cfg, 6, max_batch=B, device=0).load_params(params)
    xt, lt = torch.as_tensor(x, device="cuda:0"), torch.as_tensor(labels, device="cuda:0")
    eng.profile(True)
    got = eng.unet(xt, lt).cpu().numpy()
    seen = {p["name"] for p in eng.profile_read()}
    eng.profile(False)
    alone = eng.unet(xt[:1], lt[:1]).cpu().numpy()
    eng.close()
    err = np.abs(got[:1] - want64).max() / scale
    print("64x64 B=16: |hip-f64| %.3e (float32 oracle %.3e), batch vs alone %.3e" % (err, e_f32, np.abs(got[:1] - alone).max() / scale))
    # the 64x64 and 32x32 levels have enough tiles (>= 256 workgroup tiles) for the hybrid Winograd kernel
    assert "conv3x3_wino42_f32_kernel" in seen, seen
    assert np.isfinite(got).all() and err < max(4 * e_f32, 2e-5), (err, e_f32)
    assert np.abs(got[:1] - alone).max() / scale < 2e-5


@pytest.mark.parametrize("kernel", ["wino42", "wino42_n32", "wino2d"])
def test_epilogue_groupnorm_16x16_level(native_lib, tmp_path, kernel):
    """GroupNorm in the producer's epilogue (plan.h: Op::egn_peer) on 16x16 images, where the image is one workgroup tile of the
    Winograd kernel -- the hybrid F(2,3)y x F(4,3)x kernel of round 3 (persistent form for Cin = 64 with its 64-output and, round 4, its
    32-output workgroup tile; one-tile form for Cin = 32) and the F(2x2,3x3) kernel it replaced (kept as the fallback) -- and the column sums of its waves are added through LDS: a three-level network
    run with the kernel forced on at a small batch (the switches are read once per process, hence the worker), with and without the
    fusion.  Both runs match the float64 oracle, the fused one launches fewer GroupNorm passes, and the two agree with each other."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    res = {}
    for tag, extra in (("egn", {}), ("noegn", {"GDDIM_NO_EGN": "1"})):
        out = str(tmp_path / (tag + ".npz"))
        force = {"wino42": {"GDDIM_WINO42_MINBLOCKS": "1", "GDDIM_WINO42_NW": "2"}, "wino42_n32": {"GDDIM_WINO42_MINBLOCKS": "1", "GDDIM_WINO42_NW": "1"},
                 "wino2d": {"GDDIM_NOWINO42": "1", "GDDIM_WINO2_MINBLOCKS": "1"}}[kernel]
        env = dict(os.environ, **force, **extra)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "egn_worker.py"), out], cwd=ROOT, env=env, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0 and "EGN_WORKER_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
        res[tag] = np.load(out)
    want = res["egn"]["want"]
    scale = max(1.0, np.abs(want).max())
    counts = {tag: dict(zip(res[tag]["names"].tolist(), res[tag]["launches"].tolist())) for tag in res}
    print("gn_apply launches: fused %d, unfused %d" % (counts["egn"].get("gn_apply", 0), counts["noegn"].get("gn_apply", 0)))
    kname = "conv3x3_%s_f32_kernel" % kernel
    assert kname in counts["egn"] and kname in counts["noegn"], counts["egn"]
    # 16x16 level: 2 + 3 residual blocks with a 3x3 producer in front of their second GroupNorm; the 8x8 level adds its own
    assert counts["egn"].get("gn_apply", 0) <= counts["noegn"]["gn_apply"] - 5
    for tag in res:
        err = np.abs(res[tag]["got"] - want).max()
        assert np.isfinite(res[tag]["got"]).all() and err < 1e-4 * scale, (tag, err, scale)
    assert np.abs(res["egn"]["got"] - res["noegn"]["got"]).max() < 2e-5 * scale


def test_unet_agrees_across_batch_sizes(native_lib):
    """The kernel a layer runs on depends on the batch size (direct implicit GEMM, the hybrid Winograd kernel's two tiles, F(2x2,3x3)'s
    channel-split launch, the epilogue GroupNorm).  Every batch is built from 4 distinct (image, label) pairs laid over ALL its positions
    (seeded random assignment): inside a batch every copy of a pair must agree bit for bit wherever it sits (only out[:2] used to
    be read, positions 2 .. B-1 were checked for finiteness), and the distinct ones agree with their B = 4 evaluation to fp32 round-off
    across batches of 1 .. 320."""
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.engine import Engine
    cfg = cld_accr_dcifar10.get_config()
    params = onet.random_params(cfg, 6, seed=3)
    eng = Engine(cfg, 6, max_batch=320, device=0)
    eng.load_params(params)
    rng = np.random.default_rng(5)
    D = 4
    xd = torch.as_tensor(rng.standard_normal((D, 32, 32, 6)).astype(np.float32), device="cuda:0")
    labd = torch.as_tensor((rng.uniform(1e-3, 1, D) * 999).astype(np.float32), device="cuda:0")
    ref = eng.unet(xd, labd).cpu().numpy()
    scale = np.abs(ref).max()
    for B in (1, 5, 16, 31, 64, 65, 128, 200, 256, 320):
        lay = np.arange(B) % D
        np.random.default_rng(B).shuffle(lay)
        idx = torch.as_tensor(lay, device="cuda:0", dtype=torch.long)
        out = eng.unet(xd[idx].contiguous(), labd[idx].contiguous()).cpu().numpy()
        assert np.isfinite(out).all()
        first = {}
        for p, j in enumerate(lay):
            first.setdefault(int(j), p)
        bad = [p for p, j in enumerate(lay) if not np.array_equal(out[p], out[first[int(j)]])]
        assert not bad, (B, len(bad), bad[:4])
        err = max(np.abs(out[p] - ref[j]).max() for j, p in first.items()) / scale
        assert err < 2e-5, (B, err)
    eng.close()


def test_3x3_levels_stay_on_the_hybrid_kernel_across_batch_sizes(native_lib, gd_env):
    """at the reference's own per-device batch (eval.batch_size / 8 = 128, cld_jax/run_lib.py:721) the 8x8 level has 128
    wide workgroup tiles and used to fall off the hybrid Winograd kernel onto the direct one (105 instead of ~300 TFLOP/s algorithmic), the
    16x16 level likewise below B = 64.  With the 32-output tile the 3x3 stride-1 convolutions of the 32x32 .. 8x8 levels run the hybrid kernel
    at EVERY batch size -- the narrow tile where the wide one would leave CUs idle, the wide one at the production batch -- and the 4x4 level
    its channel-split F(2x2,3x3) launch.  Kernel per level from the engine's profile ("label class|kernel" keys)."""
    import re
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.engine import Engine
    gd_env(GD_PROF_BY_LABEL="2")
    cfg = cld_accr_dcifar10.get_config()
    params = onet.random_params(cfg, 6, seed=3)
    eng = Engine(cfg, 6, max_batch=256, device=0)
    eng.load_params(params)
    rng = np.random.default_rng(6)
    x = torch.as_tensor(rng.standard_normal((256, 32, 32, 6)).astype(np.float32), device="cuda:0")
    lab = torch.full((256,), 500.0, device="cuda:0")
    want = {16: {32: "n32", 16: "n32", 8: "n32"}, 64: {32: "wide", 16: "wide", 8: "n32"}, 128: {32: "wide", 16: "wide", 8: "n32"},
            256: {32: "wide", 16: "wide", 8: "wide"}}
    for B, levels in want.items():
        eng.profile(True)
        eng.unet(x[:B], lab[:B])
        prof = eng.profile_read()
        eng.profile(False)
        seen = {}
        for p in prof:
            # label classes have their digits removed: Conv_0 / Conv_1 (3x3) and the 1x1 shortcut Conv_2 of a residual block all read
            # "ResnetBlockBigGANpp_.conv@HxW:Cin>Cout"; the 3x3 ones are those on a conv3x3_* kernel -- or, if the dispatch regressed, the
            # launches missing from the per-level counts below (36 / 38 / 38 / 40 at 32x32 / 16x16 / 8x8 / 4x4)
            m = re.match(r"ResnetBlockBigGANpp_\.conv@(\d+)x\d+:\d+>\d+\|(conv3x3_\w+)", p["name"])
            if m:
                d = seen.setdefault(int(m.group(1)), {})
                d[m.group(2)] = d.get(m.group(2), 0) + p["launches"]
        counts = {32: 36, 16: 38, 8: 38, 4: 40}
        for H, tile in levels.items():
            kern = "conv3x3_wino42_n32_f32_kernel" if tile == "n32" else "conv3x3_wino42_f32_kernel"
            got = seen.get(H, {})
            assert all(k.startswith("conv3x3_wino42") for k in got) and sum(got.values()) == counts[H], (B, H, got)
            assert max(got, key=got.get) == kern, (B, H, got)
        assert seen.get(4) == {"conv3x3_wino2d_splitk_f32_kernel": counts[4]}, (B, seen.get(4))
    eng.close()


def test_groupnorm_side_output_matches_standalone_pass(native_lib, gd_env):
    """The up-path blocks' first GroupNorm (+ swish) is written by the shortcut convolution's loader (`igemm_f32_pers_kernel<...>+gn`,
    Op::side_peer) at batches where that launch is persistent: the same operations with the same roundings as `gn_apply_kernel`, so the
    network output must equal -- to fp32 round-off of the whole network, in practice bit for bit -- the engine built with
    GDDIM_NO_SIDE_GN=1, where the stand-alone pass does it.  Two engines under two option snapshots in one process (csrc/options.h).
    (Round 5 tried the side output spread over the N tiles and in packed fp32: no time gained, and a hand-written packed multiply right
    behind a v_rcp read a stale register -- this test and the batch sweep above caught it at the first batch on this path.)"""
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.engine import Engine
    cfg = cld_accr_dcifar10.get_config()
    params = onet.random_params(cfg, 6, seed=9)
    rng = np.random.default_rng(10)
    B = 128
    x = torch.as_tensor(rng.standard_normal((B, 32, 32, 6)).astype(np.float32), device="cuda:0")
    lab = torch.as_tensor((rng.uniform(1e-3, 1, B) * 999).astype(np.float32), device="cuda:0")
    outs, names = {}, {}
    for tag, off in (("side", None), ("standalone", "1")):
        gd_env(GDDIM_NO_SIDE_GN=off)
        eng = Engine(cfg, 6, max_batch=B, device=0)
        eng.load_params(params)
        eng.profile(True)
        outs[tag] = eng.unet(x, lab).cpu().numpy()
        names[tag] = [p["name"] for p in eng.profile_read()]
        eng.profile(False)
        eng.close()
    assert any(n.endswith("+gn") for n in names["side"]) and not any(n.endswith("+gn") for n in names["standalone"]), names["side"]
    scale = np.abs(outs["standalone"]).max()
    err = np.abs(outs["side"] - outs["standalone"]).max() / scale
    print("GroupNorm side output vs stand-alone pass: rel max|d| %.3e, bit-identical: %s" % (err, np.array_equal(outs["side"], outs["standalone"])))
    assert np.isfinite(outs["side"]).all() and err < 2e-6, err
