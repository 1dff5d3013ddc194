"""CPU checks of the C-ABI boundary: the library loads without a GPU, exports every symbol include/gddim_amd.h
declares, plans on the host, and refuses -- loudly -- to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, tiny_config


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "gddim_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gd_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(native_lib):
    from gddim_amd import _abi
    declared = _declared_symbols()
    assert len(declared) >= 25
    raw = ctypes.CDLL(_abi.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), "symbol %s declared in include/gddim_amd.h is not exported" % name
    assert sorted(_abi.SIGNATURES) == declared, "ctypes signature table and header disagree"
    assert native_lib.gd_version() >= 100


def test_plan_only_engine_and_param_errors(native_lib):
    from gddim_amd._abi import GddimError
    from gddim_amd.engine import Engine
    cfg = tiny_config("cld")
    eng = Engine(cfg, 6, max_batch=2, plan_only=True)
    specs = eng.param_specs()
    assert "ResnetBlockBigGANpp_0/Conv_0/kernel" in specs and specs["Conv_0/kernel"] == (3, 3, 6, 16)
    assert specs["Downsample_0/Conv2d_0/weight"] == (3, 3, 6, 16)
    plan = eng.plan()
    assert plan["CP"] == 8 and plan["arena_per_image"] > 0
    with pytest.raises(GddimError):  # unknown parameter
        eng.load_params({"nope": np.zeros(3, np.float32)})
    with pytest.raises(GddimError):  # wrong shape
        eng.load_params({"Conv_0": {"kernel": np.zeros((3, 3, 5, 16), np.float32)}})
    with pytest.raises(GddimError):  # incomplete tree
        eng.load_params({"Conv_0": {"bias": np.zeros(16, np.float32)}})
    eng.close()


def test_unsupported_architecture_is_rejected(native_lib):
    from gddim_amd._abi import GddimError
    from gddim_amd.engine import Engine
    cfg = tiny_config("cld")
    cfg.model.resblock_type = 'ddpm'
    with pytest.raises(GddimError):
        Engine(cfg, 6, 1, plan_only=True)
    cfg = tiny_config("cld")
    cfg.model.nf = 24  # GroupNorm groups of 6 channels: not a multiple of 16
    with pytest.raises(GddimError):
        Engine(cfg, 6, 1, plan_only=True)


def test_no_cpu_fallback(native_lib):
    """Without a GPU every compute entry point fails; nothing routes through the oracle or torch-CPU."""
    import torch
    from gddim_amd import ops
    from gddim_amd._abi import GddimError
    from gddim_amd.engine import Engine
    with pytest.raises(GddimError):
        ops.sbmm(np.eye(2, dtype=np.float32), torch.zeros(1, 2, 2))
    with pytest.raises(GddimError):
        ops.dct2d_32(torch.zeros(1, 32, 32, 3))
    if not torch.cuda.is_available():
        with pytest.raises(GddimError):
            Engine(tiny_config("cld"), 6, 1, device=0)
    # the product package never imports the oracle
    import gddim_amd
    pkg = os.path.dirname(gddim_amd.__file__)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), "%s imports the oracle" % f


def test_config_dict_and_factories():
    from gddim_amd.configs import cld_accr_dcifar10, blur_ddpm_deep_cifar10
    cfg = cld_accr_dcifar10.get_config().lock()
    assert cfg.sampling.method == 'deis' and cfg.model.R_dt == 1e-6 and cfg.model.is_R_rk and cfg.data.centered
    cfg.override("sampling.nfe", "50").override("sampling.deis_order", "2")
    assert cfg.sampling.nfe == 50 and cfg.sampling.deis_order == 2
    with pytest.raises(KeyError):
        cfg.override("sampling.not_a_key", "1")  # lock_config=True semantics of cld_jax/main.py:33-34
    b = blur_ddpm_deep_cifar10.get_config()
    assert b.sampling.method == 'order0' and b.sampling.t0 == 1e-5 and b.model.sigma_blur_max == 10.0
    from gddim_amd.datasets import get_data_inverse_scaler
    assert get_data_inverse_scaler(cfg)(-1.0) == 0.0 and get_data_inverse_scaler(cfg)(1.0) == 1.0
