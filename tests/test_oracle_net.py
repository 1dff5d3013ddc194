"""Known-answer tests for the oracle's network primitives (the third-party semantics restated in oracle/ncsnpp.py)
against independent plain-numpy / scipy formulations."""
import math

import numpy as np
import scipy.signal
import torch

from conftest import tiny_config
from oracle import ncsnpp as onet


def test_conv_same_is_cross_correlation_nhwc_hwio():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 5, 6, 3))
    w = rng.standard_normal((3, 3, 3, 4))
    b = rng.standard_normal(4)
    got = onet.conv_same(torch.as_tensor(x), {"kernel": w, "bias": b}).numpy()
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    want = np.zeros((2, 5, 6, 4))
    for dy in range(3):
        for dx in range(3):
            want += np.einsum("bhwi,io->bhwo", xp[:, dy:dy + 5, dx:dx + 6, :], w[dy, dx])  # no kernel flip
    assert np.abs(got - (want + b)).max() < 1e-12


def test_group_norm_flax_semantics():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 4, 4, 16)) * 3 + 1
    p = {"scale": rng.standard_normal(16), "bias": rng.standard_normal(16)}
    got = onet.group_norm(torch.as_tensor(x), p).numpy()
    G = 4  # min(16 // 4, 32)
    want = np.empty_like(x)
    for bi in range(2):
        for g in range(G):
            blk = x[bi, :, :, g * 4:(g + 1) * 4]  # contiguous channel groups
            m, v = blk.mean(), (blk * blk).mean() - blk.mean() ** 2
            want[bi, :, :, g * 4:(g + 1) * 4] = (blk - m) / np.sqrt(v + 1e-6)
    want = want * p["scale"] + p["bias"]
    assert np.abs(got - want).max() < 1e-10
    tg = torch.nn.functional.group_norm(torch.as_tensor(x).permute(0, 3, 1, 2), G, torch.as_tensor(p["scale"]),
                                        torch.as_tensor(p["bias"]), eps=1e-6).permute(0, 2, 3, 1).numpy()
    assert np.abs(got - tg).max() < 1e-10


def test_fir_resampling_matches_scipy_upfirdn_and_closed_forms():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((1, 8, 8, 2))
    xt = torch.as_tensor(x)
    k1 = np.array([1.0, 3.0, 3.0, 1.0])
    up = onet.upsample_2d(xt, [1, 3, 3, 1]).numpy()
    dn = onet.downsample_2d(xt, [1, 3, 3, 1]).numpy()
    for c in range(2):
        plane = x[0, :, :, c]
        # upsample_2d: taps [1,3,3,1]/4 per axis (gain factor^2 over a /64 kernel), pads (2,1)
        u = scipy.signal.upfirdn(k1 / 4, plane, up=2, axis=0)
        u = scipy.signal.upfirdn(k1 / 4, u, up=2, axis=1)
        assert np.abs(up[0, :, :, c] - u[1:17, 1:17]).max() < 1e-12
        d = scipy.signal.upfirdn(k1 / 8, plane, down=1, axis=0)
        d = scipy.signal.upfirdn(k1 / 8, d, down=1, axis=1)
        # downsample_2d keeps y[i] = sum_a k[a] x[2i + a - 1] = full_convolution[2i + 2]
        assert np.abs(dn[0, :, :, c] - d[2::2, 2::2][:4, :4]).max() < 1e-12
    # SURVEY.md App. E closed forms (zero boundary)
    col = x[0, :, 0, 0]
    colp = np.concatenate([[0], col, [0]])
    u1 = onet.upsample_2d(torch.as_tensor(col.reshape(1, 8, 1, 1)), [1, 3, 3, 1]).numpy()
    # 2-D kernel on a width-1 image: the W axis contributes the factor (3/4 | 3/4) at the two output columns
    for m in range(8):
        assert abs(u1[0, 2 * m, 0, 0] - 0.75 * (colp[m] + 3 * colp[m + 1]) / 4) < 1e-12
        assert abs(u1[0, 2 * m + 1, 0, 0] - 0.75 * (3 * colp[m + 1] + colp[m + 2]) / 4) < 1e-12


def test_conv_downsample_is_a_6x6_stride2_conv():
    """Appendix E: FIR pad(2,2) + 3x3 stride-2 VALID conv == 6x6 stride-2 pad-2 conv with w (*) outer([1,3,3,1])/64
    (the composition the engine packs at weight-load time)."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 8, 8, 3))
    w = rng.standard_normal((3, 3, 3, 5))
    got = onet.conv_downsample_2d(torch.as_tensor(x), torch.as_tensor(w), [1, 3, 3, 1]).numpy()
    k = np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0
    w6 = np.zeros((6, 6, 3, 5))
    for a in range(3):
        for b in range(3):
            for s in range(4):
                for t in range(4):
                    w6[a + s, b + t] += w[a, b] * k[s, t]
    want = torch.nn.functional.conv2d(torch.as_tensor(x).permute(0, 3, 1, 2), torch.as_tensor(w6).permute(3, 2, 0, 1),
                                      stride=2, padding=2).permute(0, 2, 3, 1).numpy()
    assert got.shape == (2, 4, 4, 5) and np.abs(got - want).max() < 1e-12


def test_attention_block_is_softmax_attention():
    rng = np.random.default_rng(4)
    cfg = tiny_config("cld")
    C = 16
    x = rng.standard_normal((2, 4, 4, C))
    p = {"GroupNorm_0": {"scale": np.ones(C), "bias": np.zeros(C)}}
    for i in range(4):
        p["NIN_%d" % i] = {"W": rng.standard_normal((C, C)) * 0.3, "b": rng.standard_normal(C) * 0.1}
    got = onet.attn_block(torch.as_tensor(x), p, cfg).numpy()
    h = onet.group_norm(torch.as_tensor(x), p["GroupNorm_0"]).numpy().reshape(2, 16, C)
    q, k, v = (h @ p["NIN_%d" % i]["W"] + p["NIN_%d" % i]["b"] for i in range(3))
    s = np.einsum("bnc,bmc->bnm", q, k) / math.sqrt(C)
    w = np.exp(s - s.max(-1, keepdims=True))
    w /= w.sum(-1, keepdims=True)
    o = np.einsum("bnm,bmc->bnc", w, v) @ p["NIN_3"]["W"] + p["NIN_3"]["b"]
    want = (x.reshape(2, 16, C) + o) / math.sqrt(2.0)
    assert np.abs(got.reshape(2, 16, C) - want).max() < 1e-10


def test_cld_eps_fn_channel_packing_and_labels():
    """'b ... d g -> b ... (g d)': channels are [x_0..x_{C-1}, v_0..v_{C-1}]; labels = 999 t."""
    cfg = tiny_config("cld")
    params = onet.random_params(cfg, 6, seed=0)
    rng = np.random.default_rng(5)
    u = rng.standard_normal((2, 16, 16, 3, 2))
    t = np.array([0.3, 0.7])
    eps = onet.cld_eps_fn(params, cfg, dtype=torch.float64)(u, t)
    x6 = np.concatenate([u[..., 0], u[..., 1]], axis=-1)
    with torch.no_grad():
        out6 = onet.ncsnpp_forward(params, torch.as_tensor(x6), torch.as_tensor(t * 999), cfg).numpy()
    assert np.abs(eps[..., 0] - out6[..., :3]).max() < 1e-12 and np.abs(eps[..., 1] - out6[..., 3:]).max() < 1e-12


def test_param_tree_matches_survey_counts():
    from gddim_amd.configs import cld_accr_dcifar10
    shapes = onet.param_shapes(cld_accr_dcifar10.get_config(), 6)
    assert sum(1 for k in shapes if k.startswith("ResnetBlockBigGANpp_") and k.endswith("Conv_0/kernel")) == 76
    assert sum(1 for k in shapes if k.startswith("AttnBlockpp_") and k.endswith("NIN_0/W")) == 10
    assert sum(1 for k in shapes if k.endswith("Conv_2/kernel")) == 43            # 1x1 shortcuts
    assert abs(sum(int(np.prod(s)) for s in shapes.values()) / 1e6 - 107.60) < 0.01
