"""Stand-alone HIP operators of the C ABI on torch device tensors (plumbing: pointers + stream only).

Each wrapper cites the reference helper it replaces; see include/gddim_amd.h for the contracts."""
import ctypes

from . import _abi
from ._abi import GddimError, check


def _need_gpu(t):
    if not t.is_cuda:
        raise GddimError("gddim_amd operators run on the GPU only (got a %s tensor); there is no CPU fallback" % t.device)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream(t):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _f32c(t):
    import torch
    return t.to(torch.float32).contiguous()


def ab_step(x, deis_coef, new_eps, eps_pred):
    """multistep_ab_step (cld_jax/deis.py:141-151): returns (x', full_eps[:-1])."""
    import torch
    _need_gpu(x)
    x, new_eps, eps_pred = _f32c(x), _f32c(new_eps), _f32c(eps_pred)
    coef = _f32c(torch.as_tensor(deis_coef, device=x.device))
    n_hist = eps_pred.shape[0]
    assert coef.shape[0] == n_hist + 2 and x.shape[-1] == 2
    out, hist = torch.empty_like(x), torch.empty_like(eps_pred)
    check(_abi.lib().gd_ab_step(_ptr(x), _ptr(coef), _ptr(new_eps), _ptr(eps_pred), _ptr(out), _ptr(hist), n_hist,
                                x.numel() // 2, _stream(x)))
    return out, hist


def ab_step_scalar(x, ei_coef, new_eps, eps_pred):
    """ab_step (blur_jax/multistep.py:94-97)."""
    import torch
    _need_gpu(x)
    x, new_eps, eps_pred = _f32c(x), _f32c(new_eps), _f32c(eps_pred)
    coef = _f32c(torch.as_tensor(ei_coef, device=x.device))
    n_hist = eps_pred.shape[0]
    assert coef.shape[0] == n_hist + 2
    out, hist = torch.empty_like(x), torch.empty_like(eps_pred)
    check(_abi.lib().gd_ab_step_scalar(_ptr(x), _ptr(coef), _ptr(new_eps), _ptr(eps_pred), _ptr(out), _ptr(hist), n_hist,
                                       x.numel(), _stream(x)))
    return out, hist


def sbmm(b1, b2):
    """einsum('ij,b...dj->b...di') (cld_jax/utils.py:183-185)."""
    import torch
    _need_gpu(b2)
    b2 = _f32c(b2)
    m = _f32c(torch.as_tensor(b1, device=b2.device))
    out = torch.empty_like(b2)
    check(_abi.lib().gd_bmm2x2(_ptr(m), _ptr(b2), _ptr(out), b2.numel() // 2, 0, 1, _stream(b2)))
    return out


def bmm(b1, b2):
    """einsum('bij,b...dj->b...di') (cld_jax/utils.py:179-181)."""
    import torch
    _need_gpu(b2)
    b2 = _f32c(b2)
    m = _f32c(torch.as_tensor(b1, device=b2.device))
    out = torch.empty_like(b2)
    n = b2.numel() // 2
    check(_abi.lib().gd_bmm2x2(_ptr(m), _ptr(b2), _ptr(out), n, 1, n // b2.shape[0], _stream(b2)))
    return out


def dct2d_32(x, inverse=False):
    """batch_img_dct / batch_img_idct (blur_jax/blur.py:99-106) on (B,32,32,C)."""
    import torch
    _need_gpu(x)
    x = _f32c(x)
    assert x.shape[1] == 32 and x.shape[2] == 32
    out = torch.empty_like(x)
    check(_abi.lib().gd_dct2d_32(_ptr(x), _ptr(out), x.shape[0], x.shape[3], int(inverse), _stream(x)))
    return out


def conv2d(x, w_packed, bias=None, res=None, KH=3, KW=3, stride=1, pad=1, alpha=1.0, out_scale=1.0):
    """NHWC conv with packed weights [Cout][KH][KW][Cin] through the MFMA implicit-GEMM kernel."""
    import torch
    _need_gpu(x)
    x, w_packed = _f32c(x), _f32c(w_packed)
    B, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    out = torch.empty((B, OH, OW, Cout), dtype=torch.float32, device=x.device)
    bias = _f32c(bias) if bias is not None else None
    res = _f32c(res) if res is not None else None
    check(_abi.lib().gd_conv2d(_ptr(x), _ptr(w_packed), _ptr(bias), _ptr(res), _ptr(out), B, H, W, Cin, Cout, KH, KW, stride,
                               pad, float(alpha), float(out_scale), _stream(x)))
    return out


def conv3x3_winograd2d(x, w_packed, bias=None, res=None, alpha=1.0, out_scale=1.0, u_scratch=None):
    """3x3 / stride 1 / pad 1 NHWC conv through the fp32 two-dimensional Winograd F(2x2,3x3) kernel (same packed weights as conv2d)."""
    import torch
    _need_gpu(x)
    x, w_packed = _f32c(x), _f32c(w_packed)
    B, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
    if u_scratch is None:
        u_scratch = torch.empty(Cout * 16 * Cin, dtype=torch.float32, device=x.device)
    bias = _f32c(bias) if bias is not None else None
    res = _f32c(res) if res is not None else None
    check(_abi.lib().gd_conv3x3_winograd2d(_ptr(x), _ptr(w_packed), _ptr(u_scratch), _ptr(bias), _ptr(res), _ptr(out), B, H, W, Cin,
                                         Cout, float(alpha), float(out_scale), _stream(x)))
    return out


def conv3x3_winograd42(x, w_packed, bias=None, res=None, alpha=1.0, out_scale=1.0, u_scratch=None):
    """3x3 / stride 1 / pad 1 NHWC conv through the fp32 hybrid Winograd F(2,3)_y x F(4,3)_x kernel (same packed weights as conv2d)."""
    import torch
    _need_gpu(x)
    x, w_packed = _f32c(x), _f32c(w_packed)
    B, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
    if u_scratch is None:
        u_scratch = torch.empty(24 * ((Cout + 31) // 32 * 32) * Cin, dtype=torch.float32, device=x.device)
    bias = _f32c(bias) if bias is not None else None
    res = _f32c(res) if res is not None else None
    check(_abi.lib().gd_conv3x3_winograd42(_ptr(x), _ptr(w_packed), _ptr(u_scratch), _ptr(bias), _ptr(res), _ptr(out), B, H, W, Cin,
                                           Cout, float(alpha), float(out_scale), _stream(x)))
    return out


def conv3x3_winograd2d_split6_weights(w_packed):
    """Pre-split Winograd-domain weight fragments of conv3x3_winograd2d_split6 (a uint8 device tensor; once per parameter load)."""
    import torch
    _need_gpu(w_packed)
    w_packed = _f32c(w_packed)
    Cout, Cin = w_packed.shape[0], w_packed.shape[1] // 9
    u = torch.empty(_abi.lib().gd_conv3x3_winograd2d_split6_weight_bytes(Cin, Cout), dtype=torch.uint8, device=w_packed.device)
    check(_abi.lib().gd_conv3x3_winograd2d_split6_weights(_ptr(w_packed), _ptr(u), Cin, Cout, _stream(w_packed)))
    return u


def conv3x3_winograd2d_split6(x, u_split, cout, bias=None, res=None, out_scale=1.0):
    """3x3 / stride 1 / pad 1 NHWC conv through F(2x2,3x3) in the exact three-piece bf16 split on the bf16 matrix core (the operator of precision mode 2:
    gd_conv3x3_winograd2d_split6); u_split from conv3x3_winograd2d_split6_weights."""
    import torch
    _need_gpu(x)
    x = _f32c(x)
    B, H, W, Cin = x.shape
    out = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    bias = _f32c(bias) if bias is not None else None
    res = _f32c(res) if res is not None else None
    check(_abi.lib().gd_conv3x3_winograd2d_split6(_ptr(x), _ptr(u_split), _ptr(bias), _ptr(res), _ptr(out), B, H, W, Cin, int(cout),
                                                  float(out_scale), _stream(x)))
    return out


def conv3x3_winograd2d_split6_gn(x, u_split, cout, gamma, beta, groups, act=True, bias=None, res=None, out_scale=1.0, keep_raw=True):
    """conv3x3_winograd2d_split6 followed by GroupNorm (+ swish) in the kernel's epilogue (gd_conv3x3_winograd2d_split6_gn; 16x16 images):
    returns (raw, normalised), raw = None with keep_raw=False."""
    import torch
    _need_gpu(x)
    x, gamma, beta = _f32c(x), _f32c(gamma), _f32c(beta)
    B, H, W, Cin = x.shape
    raw = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device) if keep_raw else None
    gn = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    bias = _f32c(bias) if bias is not None else None
    res = _f32c(res) if res is not None else None
    check(_abi.lib().gd_conv3x3_winograd2d_split6_gn(_ptr(x), _ptr(u_split), _ptr(bias), _ptr(res), _ptr(raw), _ptr(gamma), _ptr(beta), _ptr(gn),
                                                     int(groups), int(bool(act)), B, H, W, Cin, int(cout), float(out_scale), _stream(x)))
    return raw, gn


def conv3x3_winograd42_gn(x, w_packed, gamma, beta, groups, act=True, bias=None, res=None, out_scale=1.0, keep_raw=True):
    """conv3x3_winograd42 followed by GroupNorm (+ swish) in the kernel's epilogue (gd_conv3x3_winograd42_gn): returns (raw, normalised),
    raw = None with keep_raw=False (the engine's sole-reader case: the convolution output itself is never stored)."""
    import torch
    _need_gpu(x)
    x, w_packed, gamma, beta = _f32c(x), _f32c(w_packed), _f32c(gamma), _f32c(beta)
    B, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    raw = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device) if keep_raw else None
    gn = torch.empty((B, H, W, Cout), dtype=torch.float32, device=x.device)
    u = torch.empty(24 * ((Cout + 31) // 32 * 32) * Cin, dtype=torch.float32, device=x.device)
    bias = _f32c(bias) if bias is not None else None
    res = _f32c(res) if res is not None else None
    check(_abi.lib().gd_conv3x3_winograd42_gn(_ptr(x), _ptr(w_packed), _ptr(u), _ptr(bias), _ptr(res), _ptr(raw), _ptr(gamma), _ptr(beta), _ptr(gn),
                                              int(groups), int(bool(act)), B, H, W, Cin, Cout, 1.0, float(out_scale), _stream(x)))
    return raw, gn


def group_norm(x, scale, bias, groups, act=False):
    """flax nn.GroupNorm(eps=1e-6) (+swish) on NHWC."""
    import torch
    _need_gpu(x)
    x, scale, bias = _f32c(x), _f32c(scale), _f32c(bias)
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    out = torch.empty_like(x)
    check(_abi.lib().gd_group_norm(_ptr(x), _ptr(scale), _ptr(bias), _ptr(out), B, HW, C, groups, int(act), _stream(x)))
    return out


def fir_resample(x, up, naive=False):
    """upsample_2d / downsample_2d with k=[1,3,3,1] (up_or_down_sampling.py:333-411); naive=True: naive_upsample_2d /
    naive_downsample_2d (:76-86), the fir=False configurations."""
    import torch
    _need_gpu(x)
    x = _f32c(x)
    B, H, W, C = x.shape
    shp = (B, 2 * H, 2 * W, C) if up else (B, H // 2, W // 2, C)
    out = torch.empty(shp, dtype=torch.float32, device=x.device)
    check(_abi.lib().gd_fir_resample(_ptr(x), _ptr(out), B, H, W, C, int(bool(up)) + (2 if naive else 0), _stream(x)))
    return out


def attention(qkv):
    """softmax(q k^T C^-1/2) v on packed (B,N,3C) projections (layerspp.py:74-78)."""
    import torch
    _need_gpu(qkv)
    qkv = _f32c(qkv)
    B, N, C3 = qkv.shape
    C = C3 // 3
    scratch = torch.empty((B, N, N), dtype=torch.float32, device=qkv.device)
    out = torch.empty((B, N, C), dtype=torch.float32, device=qkv.device)
    check(_abi.lib().gd_attention(_ptr(qkv), _ptr(scratch), _ptr(out), B, N, C, _stream(qkv)))
    return out


def attention_fused(qk, vT, bias_v=None):
    """softmax(q k^T C^-1/2) v (+ b_v) in one launch: qk (B,N,2C) packed projections, vT (B,C,N) the value projection transposed
    per image (what the engine's attention block emits); N = C = 256 only (gd_attention_fused)."""
    import torch
    _need_gpu(qk)
    qk, vT = _f32c(qk), _f32c(vT)
    B, N, C2 = qk.shape
    C = C2 // 2
    out = torch.empty((B, N, C), dtype=torch.float32, device=qk.device)
    bias_v = _f32c(bias_v) if bias_v is not None else None
    check(_abi.lib().gd_attention_fused(_ptr(qk), _ptr(vT), _ptr(bias_v), _ptr(out), B, N, C, _stream(qk)))
    return out


def attention_block_fused(q, k, vT, bias=None, res=None, out_scale=1.0, want_stats=False):
    """(res + softmax(q k^T C^-1/2) v + bias) * out_scale in one launch: q, k, res (B,N,C), vT (B,C,N); with want_stats also the fused
    GroupNorm statistics (B, N/32, C/4, 2) of the result.  N = C = 256 only (gd_attention_block_fused)."""
    import torch
    _need_gpu(q)
    q, k, vT = _f32c(q), _f32c(k), _f32c(vT)
    B, N, C = q.shape
    out = torch.empty((B, N, C), dtype=torch.float32, device=q.device)
    bias = _f32c(bias) if bias is not None else None
    res = _f32c(res) if res is not None else None
    qs = torch.empty((B, N // 32, C // 4, 2), dtype=torch.float32, device=q.device) if want_stats else None
    check(_abi.lib().gd_attention_block_fused(_ptr(q), _ptr(k), _ptr(vT), _ptr(bias), _ptr(res), float(out_scale), _ptr(qs), _ptr(out),
                                              B, N, C, _stream(q)))
    return (out, qs) if want_stats else out


def randn_pairs(n_pairs, seed, draw=0, elem0=0, device=None):
    """(n_pairs, 2) standard normals from the engine's counter-based generator (gd_randn_pairs)."""
    import torch
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else device
    out = torch.empty((int(n_pairs), 2), dtype=torch.float32, device=device)
    check(_abi.lib().gd_randn_pairs(_ptr(out), int(n_pairs), int(seed) & 0xFFFFFFFFFFFFFFFF, int(draw), int(elem0), _stream(out)))
    return out
