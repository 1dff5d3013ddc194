import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _cap_torch_threads():
    """The oracle's torch-CPU convolutions collapse when a 256-thread host pool (the GPU box) is thrown at 1-2 image problems
    (measured in round 1: 658 s instead of 6 s for the same sample); cap the pool like bench.py's cpu_baseline does."""
    try:
        import torch
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
    except ImportError:
        pass
    yield


@pytest.fixture(scope="session")
def native_lib():
    """Build (if stale) and load the native library; the build is a cross-compile and needs no GPU."""
    from gddim_amd import build, _abi
    build.build()
    return _abi.lib()


@pytest.fixture
def gd_env(monkeypatch):
    """Switches of the native library for ONE test: gd_env(GDDIM_X="1", ...) writes the environment and has the library parse it again
    (csrc/options.h: one table, read when the library first needs it, at engine creation and on gd_options_reload); undone afterwards."""
    from gddim_amd import _abi

    def set_(**kw):
        for k, v in kw.items():
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, str(v))
        _abi.reload_options()

    yield set_
    monkeypatch.undo()
    _abi.reload_options()


def tiny_config(kind="cld", nf=16, num_res_blocks=1, image_size=16, ch_mult=(1, 2, 2), attn=(8,), centered=True,
                progressive_input='residual'):
    """A small NCSN++ of the benchmarked family (biggan blocks, fir, fourier, skip_rescale)."""
    if kind == "cld":
        from gddim_amd.configs import cld_accr_dcifar10 as c
    else:
        from gddim_amd.configs import blur_ddpm_deep_cifar10 as c
    cfg = c.get_config()
    cfg.model.nf = nf
    cfg.model.num_res_blocks = num_res_blocks
    cfg.model.ch_mult = tuple(ch_mult)
    cfg.model.attn_resolutions = tuple(attn)
    cfg.model.progressive_input = progressive_input
    cfg.data.image_size = image_size
    cfg.data.centered = centered
    return cfg


def oracle_cld_like(sde, **kw):
    """The oracle's CLD in the arithmetic the product object is configured for: float32 tables for config.model.x64=False (the
    reference's default), float64 for x64=True -- so that a comparison measures the HIP path, not the tables' precision (that term is
    held by tests/test_oracle_cld.py::test_parity_budget_float32_tables_end_to_end)."""
    import numpy as np
    from oracle import cld as ocld
    return ocld.CLD(m_inv=sde.m_inv, beta_0=sde.beta_0, beta_1=sde.beta_1, vv_gamma=sde.vv_gamma, numerical_eps=sde.numerical_eps,
                    is_R_rk=sde.is_R_rk, R_dt=sde.R_dt, dtype=np.float64 if sde.x64 else np.float32, **kw)
