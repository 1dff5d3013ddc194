"""CPU checks of the native planner + weight packing: the engine's plan, executed op by op by the torch
emulation in plan_interp.py on the engine's own packed weights, must reproduce the oracle U-Net."""
import numpy as np
import pytest
import torch

from conftest import tiny_config
from oracle import ncsnpp as oracle_net
from plan_interp import PlanInterp


def _run(cfg, in_ch, B=2, seed=0):
    from gddim_amd.engine import Engine
    eng = Engine(cfg, in_ch, max_batch=B, plan_only=True)
    params = oracle_net.random_params(cfg, in_ch, seed=seed, dtype=np.float32)
    assert set(oracle_net.flatten(params)) == set(eng.param_specs()), "parameter names differ from the oracle's flax tree"
    for k, shp in eng.param_specs().items():
        assert tuple(oracle_net.flatten(params)[k].shape) == shp, k
    eng.load_params(params)
    rng = np.random.default_rng(seed + 1)
    H = cfg.data.image_size
    x = torch.as_tensor(rng.standard_normal((B, H, H, in_ch)), dtype=torch.float64)
    if not cfg.data.centered:
        x = x * 0.25 + 0.5
    labels = torch.as_tensor(rng.uniform(1.0, 999.0, size=(B,)), dtype=torch.float64)
    CP = eng.plan()["CP"]
    xin = torch.zeros(B, H, H, CP, dtype=torch.float64)
    xin[..., :in_ch] = x
    got = PlanInterp(eng).run(xin, labels)
    with torch.no_grad():
        want = oracle_net.ncsnpp_forward(params, x, labels, cfg)
    eng.close()
    return got, want


@pytest.mark.parametrize("kind,in_ch", [("cld", 6), ("blur", 3)])
def test_plan_matches_oracle_tiny(native_lib, kind, in_ch):
    got, want = _run(tiny_config(kind), in_ch)
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    # float32-packed weights (incl. the precomposed 6x6 FIR conv) vs float64 oracle on float32 weights
    assert err < 2e-5 * max(1.0, want.abs().max().item()), err


def test_plan_matches_oracle_variants(native_lib):
    # not centered (x <- 2x-1), no input pyramid, attention at two resolutions, 2 res blocks, wider ch_mult
    cfg = tiny_config("cld", nf=16, num_res_blocks=2, image_size=16, ch_mult=(1, 2, 3), attn=(8, 4), centered=False,
                      progressive_input='none')
    got, want = _run(cfg, 6, B=3, seed=3)
    err = (got - want).abs().max().item()
    assert err < 2e-5 * max(1.0, want.abs().max().item()), err


def test_plan_standalone_statistics_fallback(native_lib):
    """8x8 images: the 2x2 level has 4 rows per image, which the fused epilogue statistics do not cover (they need 16 or a
    multiple of 32 rows per image) -> the planner falls back to the stand-alone statistics kernel there."""
    from gddim_amd.engine import Engine
    cfg = tiny_config("cld", nf=16, num_res_blocks=1, image_size=8, ch_mult=(1, 2, 2), attn=(4,))
    eng = Engine(cfg, 6, max_batch=1, plan_only=True)
    kinds = [op["kind"] for op in eng.plan()["ops"]]
    eng.close()
    assert 0 < kinds.count(2) < kinds.count(3)  # some stand-alone GN_STATS ops, fewer than GroupNorms
    got, want = _run(cfg, 6, B=2, seed=7)
    assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


def test_scale_by_sigma_and_unconditional(native_lib):
    cfg = tiny_config("cld")
    cfg.model.scale_by_sigma = True
    got, want = _run(cfg, 6, B=2, seed=5)
    assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


def test_flops_match_survey(native_lib):
    """SURVEY.md App. A.3 / BASELINE.md section 2: 37.179 GFLOP per image-eval (incl. 0.011 FIR, which the engine's
    contraction count leaves out) and 33.704 GFLOP in 3x3 convolutions for the accr architecture at 32x32x6."""
    from gddim_amd.configs import cld_accr_dcifar10, blur_ddpm_deep_cifar10
    from gddim_amd.engine import Engine
    e = Engine(cld_accr_dcifar10.get_config(), 6, 1, plan_only=True)
    assert abs(e.flops_per_image() / 1e9 - (37.179 - 0.011)) < 2e-3
    assert abs(e.flops_per_image(True) / 1e9 - 33.704) < 1e-3
    assert len(e.param_specs()) == 963
    n_params = sum(int(np.prod(s)) for s in e.param_specs().values())
    assert abs(n_params / 1e6 - 107.60) < 0.01
    e.close()
    cfg = cld_accr_dcifar10.get_config()
    cfg.data.image_size = 64
    e = Engine(cfg, 6, 1, plan_only=True)
    assert abs(e.flops_per_image(True) / 1e9 - 134.815) < 1e-3
    e.close()
    e = Engine(blur_ddpm_deep_cifar10.get_config(), 3, 1, plan_only=True)
    assert abs(e.flops_per_image(True) / 1e9 - 33.688) < 1e-3
    e.close()


def test_plan_matches_oracle_ddpmpp_style(native_lib):
    """The reference's smaller configurations (configs/ddpmpp_cifar10_config.py, simple_, debug_): sinusoidal positional
    embedding (layers.py:450-464), fir=False (nearest-neighbour / 2x2-mean resampling, up_or_down_sampling.py:76-86), no input
    pyramid, 4 res blocks per level in the originals (2 here)."""
    cfg = tiny_config("cld", nf=32, num_res_blocks=2, image_size=16, ch_mult=(1, 2, 2), attn=(8,), progressive_input='none')
    cfg.model.fir = False
    cfg.model.embedding_type = 'positional'
    got, want = _run(cfg, 6, B=2, seed=5)
    err = (got - want).abs().max().item()
    assert torch.isfinite(got).all() and err < 2e-5 * max(1.0, want.abs().max().item()), err
    # blur flavour: 3 channels
    bcfg = tiny_config("blur", nf=16, num_res_blocks=1, image_size=32, ch_mult=(1, 2), attn=(16,), progressive_input='none')
    bcfg.model.fir = False
    bcfg.model.embedding_type = 'positional'
    got, want = _run(bcfg, 3, B=1, seed=6)
    assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())
    # fir=False together with the input pyramid is not a configuration of the reference: refused loudly
    from gddim_amd._abi import GddimError
    from gddim_amd.engine import Engine
    bad = tiny_config("cld")
    bad.model.fir = False
    with pytest.raises(GddimError):
        Engine(bad, 6, max_batch=1, plan_only=True)


def test_plan_groups_that_cut_float4_columns(native_lib):
    """nf = 64 (configs/debug_cifar10_config.py): the up path normalises 128 + 64 = 192 concatenated channels in 32 groups of 6,
    which the quad statistics cannot express -> the planner's generic GroupNorm op."""
    from gddim_amd.engine import Engine
    cfg = tiny_config("cld", nf=64, num_res_blocks=1, image_size=8, ch_mult=(1, 2), attn=(4,), progressive_input='none')
    cfg.model.fir = False
    cfg.model.embedding_type = 'positional'
    eng = Engine(cfg, 6, max_batch=1, plan_only=True)
    assert [op["kind"] for op in eng.plan()["ops"]].count(9) >= 1
    eng.close()
    got, want = _run(cfg, 6, B=2, seed=9)
    assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


def test_every_shipped_configuration_plans(native_lib):
    """gddim_amd.configs.variants: every model configuration of the reference (CIFAR-10) lowers to a plan whose parameter tree
    has the flax names / shapes the oracle derives for it, with the expected size."""
    from gddim_amd.configs import variants
    from gddim_amd.engine import Engine
    sizes = {}
    for name in variants.names():
        cfg = variants.get_config(name)
        in_ch = 6 if name.startswith("cld") else 3
        eng = Engine(cfg, in_ch, max_batch=1, plan_only=True)
        specs = eng.param_specs()
        want = oracle_net.param_shapes(cfg, in_ch)
        assert set(specs) == set(want) and all(tuple(want[k]) == specs[k] for k in specs), name
        sizes[name] = sum(int(np.prod(s)) for s in specs.values())
        eng.close()
    assert 107e6 < sizes['cld_accr_dcifar10'] < 108e6 and sizes['cld_deep_cifar10'] == sizes['cld_accr_dcifar10']
    assert 61e6 < sizes['cld_ddpmpp_cifar10'] < 62.5e6  # the DDPM++ of Song et al. (61.8 M at 3 channels)
    assert sizes['cld_simple_cifar10'] < sizes['blur_debug_cifar10'] < sizes['blur_ddpmpp_cifar10']
    assert variants.get_config('cld_ndeep_cifar10').model.mixed_score is True


def test_epilogue_groupnorm_links(native_lib):
    """GroupNorm passes the producing GEMM can absorb (plan.h: Op::egn_peer): single-source GroupNorms of the 16x16, 8x8 and 4x4 levels (a
    32x32 image spans four workgroup tiles: built and measured null in round 3, removed in round 4).  Links are symmetric, the producer
    emits the tensor the GroupNorm reads, and `egn_sole` is set exactly when nothing else reads it."""
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.engine import Engine
    eng = Engine(cld_accr_dcifar10.get_config(), 6, max_batch=1, plan_only=True)
    ops = eng.plan()["ops"]
    eng.close()
    gns = [i for i, o in enumerate(ops) if o["kind"] == 3]
    linked = [i for i in gns if ops[i]["egn_peer"] >= 0]
    assert len(gns) == 163 and len(linked) >= 40
    for i in linked:
        a, g = ops[i], ops[ops[i]["egn_peer"]]
        assert g["kind"] == 1 and g["egn_peer"] == i and g["out"] == a["a0"] and a["C1"] == 0 and a["H"] * a["W"] in (16, 64, 256)
        assert g["N"] == a["C0"] and g["m_per_image"] == a["H"] * a["W"] and g["qstats"][0] >= 0
        t = a["a0"][0]
        others = [k for k, o in enumerate(ops) if k != i and t in (o["a0"][0], o["a1"][0], o["bten"][0], o["res"][0], o["rowbias"][0])]
        assert bool(g["egn_sole"]) == (not others)
    # the GroupNorms in front of the 16x16 attention blocks and 1x1 shortcuts are produced by GEMMs too, but only a 3x3 producer on a
    # Winograd kernel normalises them (run time); the two-source GroupNorms of the up path are never candidates
    assert all(ops[i]["egn_peer"] < 0 for i in gns if ops[i]["H"] >= 32 or ops[i]["C1"] > 0)
    assert sum(1 for i in linked if ops[i]["H"] == 16) >= 20


def _live_intervals(plan):
    """[first write, last access] (op indices) of every arena tensor.  A GroupNorm output that its producer's epilogue (egn_peer) or
    the block's shortcut convolution (side_peer) may write EARLY starts living at that op."""
    first, last = {}, {}
    # the attention core runs its qk / softmax / av ops as ONE launch (attn256_f32_kernel): every tensor one of the three touches is
    # live for the whole group -- a workgroup's epilogue stores `out` rows while another still reads q
    group = {}
    labels = [op["label"] for op in plan["ops"]]
    for k, lab in enumerate(labels):
        if lab.endswith(".qk") and k + 2 < len(labels) and labels[k + 1].endswith(".softmax") and labels[k + 2].endswith(".av"):
            group[k] = group[k + 1] = group[k + 2] = (k, k + 2)

    def touch(t, k, write=False):
        if k in group and t >= 0:
            lo, hi = group[k]
            if write:
                first[t] = min(first.get(t, lo), lo)
            first.setdefault(t, lo)
            last[t] = max(last.get(t, hi), hi)
            return
        _touch(t, k, write)

    def _touch(t, k, write=False):
        if t < 0:
            return
        if write:
            first[t] = min(first.get(t, k), k)
        first.setdefault(t, k)
        last[t] = max(last.get(t, k), k)

    for k, op in enumerate(plan["ops"]):
        for key in ("a0", "a1", "bten", "rowbias", "res"):
            touch(op[key][0], k)
        if op["kind"] == 3:  # GN_APPLY reads its statistics
            touch(op["stats"][0], k)
            touch(op["stats1"][0], k)
            early = min([p for p in (op["egn_peer"], op["side_peer"]) if p >= 0] + [k])
            touch(op["out"][0], early, write=True)
            touch(op["out"][0], k)
        else:
            touch(op["stats"][0], k, write=True)   # GN_STATS / GN_GENERIC write theirs
            touch(op["out"][0], k, write=True)
            touch(op["qstats"][0], k, write=True)
    return first, last


@pytest.mark.parametrize("which", ["accr", "accr64", "deep64", "tiny", "variant"])
def test_no_live_tensors_alias_including_early_epilogue_writes(native_lib, which):
    """The arena planner recycles blocks; a GroupNorm output that the producing GEMM's epilogue writes early (Op::egn_peer) may only sit
    in a block that has been free since before that GEMM (plan.cpp: arena_alloc_fresh, round 3: recycled instead of always growing the
    arena).  Check the whole plan: no two tensors whose address ranges overlap are alive at the same time."""
    from gddim_amd.configs import cld_accr_dcifar10
    from gddim_amd.engine import Engine
    from gddim_amd.configs import variants
    cfg = {"accr": cld_accr_dcifar10.get_config(), "accr64": cld_accr_dcifar10.get_config(), "deep64": variants.get_config("cld_deep_cifar10"),
           "tiny": tiny_config("cld"),
           "variant": tiny_config("cld", nf=16, num_res_blocks=2, image_size=32, ch_mult=(1, 2, 2), attn=(16,))}[which]
    if which.endswith("64"):
        cfg.data.image_size = 64  # BASELINE config 5: here first-fit used to put AttnBlockpp_k.out two rows below .qf
    eng = Engine(cfg, 6, max_batch=1, plan_only=True)
    plan = eng.plan()
    eng.close()
    first, last = _live_intervals(plan)
    T = plan["tensors"]
    live = [(t, T[t]["offset"], T[t]["offset"] + (T[t]["per_image"] + 63) // 64 * 64, first[t], last[t])
            for t in first if T[t]["ext"] == 0 and T[t]["offset"] >= 0]
    live.sort(key=lambda r: r[1])
    bad = []
    for i, (ta, a0, a1, fa, la) in enumerate(live):
        for tb, b0, b1, fb, lb in live[i + 1:]:
            if b0 >= a1:
                break
            if fa <= lb and fb <= la:  # address ranges overlap and so do the lifetimes
                bad.append((T[ta]["name"], (fa, la), T[tb]["name"], (fb, lb)))
    assert not bad, bad[:5]
    assert plan["arena_per_image"] <= max(r[2] for r in live)
    if which == "accr":
        # 4.51 M floats per image in round 2 (every early-written block fresh), 2.79 M before the epilogue GroupNorm existed, 3.70 M with
        # recycled early blocks
        assert plan["arena_per_image"] < 3.8e6, plan["arena_per_image"]


@pytest.mark.parametrize("switch", ["GDDIM_NO_ATTN_FOLD", "GDDIM_NO_UPSWAP"])
def test_written_forms_behind_the_lowering_switches_match_oracle(native_lib, gd_env, switch):
    """The planner's two algebraic rewrites -- attention blocks with two of their four projections folded away (Plan::attn), up-sampling
    blocks multiplying their 1x1 shortcut before the FIR (Plan::resblock) -- have the written forms behind switches that are read per
    plan (csrc/options.h; ADVICE r3: they used to be latched in function-local statics on the first plan of the process, and nothing ran
    them).  Both forms of both rewrites, built in ONE process, executed by the float64 plan interpreter on the engine's packed weights,
    against the oracle's literal blocks; the switch changes the program (op counts) and not the function."""
    from gddim_amd.engine import Engine
    cfg = tiny_config("cld", nf=16, num_res_blocks=2, image_size=16, ch_mult=(1, 2, 2), attn=(8,))
    counts = {}
    for on in (False, True):
        gd_env(**{switch: "1" if on else None})
        eng = Engine(cfg, 6, max_batch=1, plan_only=True)
        plan = eng.plan()
        eng.close()
        assert plan["options"][switch] == (1 if on else 0)
        labels = [op["label"] for op in plan["ops"]]
        counts[on] = {"ops": len(labels), "q_proj": sum(l.endswith(".q_proj") for l in labels), "qk_proj": sum(l.endswith(".qk_proj") for l in labels),
                      "proj": sum(l.startswith("AttnBlockpp") and l.endswith(".proj") for l in labels)}
        got, want = _run(cfg, 6, B=2, seed=11)
        err = (got - want).abs().max().item()
        assert torch.isfinite(got).all() and err < 2e-5 * max(1.0, want.abs().max().item()), (switch, on, err)
    if switch == "GDDIM_NO_ATTN_FOLD":
        assert counts[False]["q_proj"] > 0 and counts[False]["proj"] == 0 and counts[True]["qk_proj"] == counts[False]["q_proj"] and counts[True]["proj"] > 0
    assert counts[True]["ops"] != counts[False]["ops"] or switch == "GDDIM_NO_UPSWAP"


def test_engines_keep_the_option_table_they_were_created_under(native_lib, gd_env):
    """the option table used to be one process-wide global that every gd_engine_create / gd_options_reload
    overwrote, while selectors read it live -- an engine built under one setting ran under the next one's.  Now every engine keeps the
    immutable snapshot of its creation and installs it for the duration of its calls; a reload (or another engine's creation) never
    changes a live engine.  Also the parsing rules: a switch set to a word is on unless the word says off; a numeric entry that does not
    parse is ignored; an outer switch forces the switches nested under it (GDDIM_NOWINO = no Winograd at all, whatever GDDIM_NOWINO42 says: ADVICE r5)."""
    from gddim_amd.engine import Engine
    cfg = tiny_config("cld", nf=16, num_res_blocks=1, image_size=16, ch_mult=(1, 2), attn=(8,))
    gd_env(GDDIM_NO_ATTN_FOLD=None, GDDIM_NO_EGN=None)
    a = Engine(cfg, 6, max_batch=1, plan_only=True)
    gd_env(GDDIM_NO_ATTN_FOLD="1", GDDIM_NO_EGN="yes", GDDIM_WINO42_GRID="3")
    b = Engine(cfg, 6, max_batch=1, plan_only=True)
    gd_env(GDDIM_NO_ATTN_FOLD=None, GDDIM_NO_EGN="off", GDDIM_WINO42_GRID="junk", GDDIM_IGEMM_BIGMIN="", GDDIM_NOWINO="1", GDDIM_NOWINO42="0")
    c = Engine(cfg, 6, max_batch=1, plan_only=True)
    oa, ob, oc = a.plan()["options"], b.plan()["options"], c.plan()["options"]  # read AFTER the later reloads
    assert (oa["GDDIM_NO_ATTN_FOLD"], oa["GDDIM_NO_EGN"], oa["GDDIM_WINO42_GRID"]) == (0, 0, 0)
    assert (ob["GDDIM_NO_ATTN_FOLD"], ob["GDDIM_NO_EGN"], ob["GDDIM_WINO42_GRID"]) == (1, 1, 3)
    assert (oc["GDDIM_NO_ATTN_FOLD"], oc["GDDIM_NO_EGN"], oc["GDDIM_WINO42_GRID"], oc["GDDIM_IGEMM_BIGMIN"]) == (0, 0, 0, 384)
    assert (oc["GDDIM_NOWINO"], oc["GDDIM_NOWINO2"], oc["GDDIM_NOWINO42"], oc["GDDIM_NOWINO2_SPLITK"]) == (1, 1, 1, 1)  # the outer switch wins
    # and the plans differ the way their own tables say
    la, lb = [op["label"] for op in a.plan()["ops"]], [op["label"] for op in b.plan()["ops"]]
    assert any(l.endswith(".q_proj") for l in la) and not any(l.endswith(".q_proj") for l in lb)
    for e in (a, b, c):
        e.close()


def test_precision_modes_are_selectable_without_a_gpu(native_lib, gd_env):
    """The three arithmetics of the engine (0 fp32, 1 bf16x3, 2 split6: round 6) by name, by number and through GDDIM_PRECISION at engine
    creation; an unknown mode is an error with a message, not a silent fp32."""
    from gddim_amd._abi import GddimError
    from gddim_amd.engine import Engine
    cfg = tiny_config("cld", nf=16, num_res_blocks=1, image_size=16, ch_mult=(1, 2), attn=(8,))
    for name in ("fp32", "bf16x3", "split6"):
        gd_env(GDDIM_PRECISION=name)
        e = Engine(cfg, 6, max_batch=1, plan_only=True)
        assert e.plan()["options"]["GDDIM_PRECISION"] == name
        for mode in ("fp32", "bf16x3", "split6", 0, 1, 2):
            e.set_precision(mode)
        with pytest.raises(KeyError):
            e.set_precision("fp16")
        with pytest.raises(GddimError, match="unknown precision mode"):
            from gddim_amd._abi import check
            check(e._lib.gd_engine_set_precision(e._h, 3))
        e.close()
    gd_env(GDDIM_PRECISION=None)
