"""ORACLE (test infrastructure only): NCSN++ score network forward on CPU with torch.

Restates the live branches (SURVEY.md App. A.4 lists the dead ones) of

  /root/reference/cld_jax/models/ncsnpp.py:41-243          NCSNpp.__call__
  /root/reference/cld_jax/models/layerspp.py:33-43          GaussianFourierProjection
  /root/reference/cld_jax/models/layerspp.py:61-83          AttnBlockpp
  /root/reference/cld_jax/models/layerspp.py:115-143        Downsample (fir, with_conv)
  /root/reference/cld_jax/models/layerspp.py:180-227        ResnetBlockBigGANpp
  /root/reference/cld_jax/models/layers.py:39-40,66-107     swish, ddpm_conv1x1 / ddpm_conv3x3
  /root/reference/cld_jax/models/layers.py:467-493          NIN / contract_inner
  /root/reference/cld_jax/models/up_or_down_sampling.py:40-73,168-411
                                                            Conv2d(down), conv_downsample_2d,
                                                            upfirdn_2d, upsample_2d, downsample_2d
  /root/reference/cld_jax/models/utils.py:128-182           get_model_fn / get_eps_fn (CLD packing)
  /root/reference/blur_jax/models/utils.py:104-160          blur get_eps_fn / get_yeps_fn

Third-party semantics restated (flax==0.3.1 / jax==0.2.8, sources not under /root/reference):
  nn.Conv(padding='SAME')  : cross-correlation, NHWC x HWIO, + bias
  nn.GroupNorm(num_groups) : epsilon=1e-6, stats over (H, W, C/G) per sample, groups are
                             contiguous channel blocks, var = E[x^2] - E[x]^2, affine scale/bias
  nn.Dense                 : x @ kernel + bias, kernel (in, out)
  nn.swish                 : x * sigmoid(x);   jax.nn.softmax over the last axis

Parameters are a nested dict with flax's auto-names; arrays are numpy or
torch, any float dtype -- the forward runs in the dtype of the input.
PARITY UNPINNED -- see oracle/__init__.py.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(a, like):
    return torch.as_tensor(np.asarray(a), dtype=like.dtype)


def swish(x):
    return x * torch.sigmoid(x)  # layers.py:39-40 (nn.swish)


def conv_same(x, p, like=None):
    """nn.Conv(kernel (kh,kw), strides 1, padding='SAME') on NHWC x with HWIO kernel + bias."""
    w = _t(p["kernel"], x)  # (kh, kw, I, O)
    kh = w.shape[0]
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), padding=kh // 2)
    y = y.permute(0, 2, 3, 1)
    if "bias" in p:
        y = y + _t(p["bias"], x)
    return y


def dense(x, p):
    return x @ _t(p["kernel"], x) + _t(p["bias"], x)


def nin(x, p):
    return torch.tensordot(x, _t(p["W"], x), dims=1) + _t(p["b"], x)  # layers.py:467-493


def group_norm(x, p, eps=1e-6):
    """flax nn.GroupNorm(num_groups=min(C//4, 32)) on NHWC."""
    B, H, W, C = x.shape
    G = min(C // 4, 32)
    xg = x.reshape(B, H * W, G, C // G)
    mean = xg.mean(dim=(1, 3), keepdim=True)
    mean2 = (xg * xg).mean(dim=(1, 3), keepdim=True)
    var = mean2 - mean * mean
    y = (xg - mean) * torch.rsqrt(var + eps)
    y = y.reshape(B, H, W, C)
    return y * _t(p["scale"], x) + _t(p["bias"], x)


# ---- FIR resampling (up_or_down_sampling.py) ----
def _setup_kernel(k):
    k = np.asarray(k, dtype=np.float32)  # :319-326
    if k.ndim == 1:
        k = np.outer(k, k)
    k /= np.sum(k)
    return k


def upfirdn_2d(x, k, up, down, pad0, pad1):
    """up_or_down_sampling.py:212-294 on NHWC x; k is the 2-D FIR (numpy)."""
    B, H, W, C = x.shape
    kt = torch.as_tensor(np.asarray(k), dtype=x.dtype)
    kh, kw = kt.shape
    # upsample: insert zeros AFTER each pixel (:262-264)
    if up > 1:
        z = x.new_zeros(B, H, up, W, up, C)
        z[:, :, 0, :, 0, :] = x
        x = z.reshape(B, H * up, W * up, C)
    # pad (:267-271), pads are non-negative for every call the live branches make
    x = F.pad(x, (0, 0, pad0, pad1, pad0, pad1))
    # convolve with the flipped filter (:274-285): true convolution
    xc = x.permute(0, 3, 1, 2).reshape(-1, 1, x.shape[1], x.shape[2])
    w = torch.flip(kt, dims=(0, 1))[None, None]
    y = F.conv2d(xc, w)
    y = y.reshape(B, C, y.shape[2], y.shape[3]).permute(0, 2, 3, 1)
    return y[:, ::down, ::down, :]


def upsample_2d(x, k, factor=2, gain=1):
    k = _setup_kernel(k) * (gain * (factor ** 2))  # :366
    p = k.shape[0] - factor
    return upfirdn_2d(x, k, up=factor, down=1, pad0=(p + 1) // 2 + factor - 1, pad1=p // 2)


def downsample_2d(x, k, factor=2, gain=1):
    k = _setup_kernel(k) * gain  # :403
    p = k.shape[0] - factor
    return upfirdn_2d(x, k, up=1, down=factor, pad0=(p + 1) // 2, pad1=p // 2)


def conv_downsample_2d(x, w, k, factor=2, gain=1):
    """:168-209  FIR pad, then stride-`factor` VALID conv with HWIO weights w."""
    convH = w.shape[0]
    k = _setup_kernel(k) * gain
    p = (k.shape[0] - factor) + (convH - 1)
    x = upfirdn_2d(x, k, up=1, down=1, pad0=(p + 1) // 2, pad1=p // 2)
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), stride=factor)
    return y.permute(0, 2, 3, 1)


def naive_upsample_2d(x, factor=2):
    # up_or_down_sampling.py:76-80
    N, H, W, C = x.shape
    x = x.reshape(-1, H, 1, W, 1, C)
    x = x.repeat(1, 1, factor, 1, factor, 1)  # jnp.tile on the two singleton axes
    return x.reshape(-1, H * factor, W * factor, C)


def naive_downsample_2d(x, factor=2):
    # up_or_down_sampling.py:83-86
    N, H, W, C = x.shape
    x = x.reshape(-1, H // factor, factor, W // factor, factor, C)
    return x.mean(dim=(2, 4))


def get_timestep_embedding(timesteps, embedding_dim, max_positions=10000):
    # layers.py:450-464
    half_dim = embedding_dim // 2
    emb = math.log(max_positions) / (half_dim - 1)
    emb = torch.exp(torch.arange(half_dim, dtype=torch.float32) * -emb).to(timesteps.dtype)
    emb = timesteps[:, None] * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb


# ---- blocks ----
def resnet_block_biggan(x, temb, p, cfg, out_ch=None, up=False, down=False):
    """layerspp.py:180-227 (dropout = identity at inference)."""
    B, H, W, C = x.shape
    out_ch = out_ch if out_ch else C
    fir_kernel = cfg.model.fir_kernel
    h = swish(group_norm(x, p["GroupNorm_0"]))
    if up:
        if cfg.model.fir:
            h = upsample_2d(h, fir_kernel, factor=2)
            x = upsample_2d(x, fir_kernel, factor=2)
        else:
            h = naive_upsample_2d(h, factor=2)
            x = naive_upsample_2d(x, factor=2)
    elif down:
        if cfg.model.fir:
            h = downsample_2d(h, fir_kernel, factor=2)
            x = downsample_2d(x, fir_kernel, factor=2)
        else:
            h = naive_downsample_2d(h, factor=2)
            x = naive_downsample_2d(x, factor=2)
    h = conv_same(h, p["Conv_0"])
    if temb is not None:
        h = h + dense(swish(temb), p["Dense_0"])[:, None, None, :]
    h = swish(group_norm(h, p["GroupNorm_1"]))
    h = conv_same(h, p["Conv_1"])
    if C != out_ch or up or down:
        x = conv_same(x, p["Conv_2"])  # conv1x1
    if not cfg.model.skip_rescale:
        return x + h
    return (x + h) / math.sqrt(2.0)


def attn_block(x, p, cfg):
    """layerspp.py:61-83"""
    B, H, W, C = x.shape
    h = group_norm(x, p["GroupNorm_0"])
    q = nin(h, p["NIN_0"])
    k = nin(h, p["NIN_1"])
    v = nin(h, p["NIN_2"])
    w = torch.einsum("bhwc,bHWc->bhwHW", q, k) * (int(C) ** (-0.5))
    w = w.reshape(B, H, W, H * W)
    w = torch.softmax(w, dim=-1)
    w = w.reshape(B, H, W, H, W)
    h = torch.einsum("bhwHW,bHWc->bhwc", w, v)
    h = nin(h, p["NIN_3"])
    if not cfg.model.skip_rescale:
        return x + h
    return (x + h) / math.sqrt(2.0)


def pyramid_downsample(x, p, cfg):
    """layerspp.py:134-140 -> up_or_down_sampling.Conv2d(down=True) (:40-73)."""
    w = _t(p["Conv2d_0"]["weight"], x)
    y = conv_downsample_2d(x, w, cfg.model.fir_kernel)
    return y + _t(p["Conv2d_0"]["bias"], x).reshape(1, 1, 1, -1)


def ncsnpp_forward(params, x, time_cond, cfg, taps=None):
    """ncsnpp.py:41-243.  x: (B,H,W,Cin) torch; time_cond: (B,) = 999*t.  Returns (B,H,W,Cin)."""
    m = cfg.model
    nf, ch_mult, nrb = m.nf, tuple(m.ch_mult), m.num_res_blocks
    attn_res = tuple(m.attn_resolutions)
    nres = len(ch_mult)
    assert m.resblock_type.lower() == "biggan" and m.embedding_type.lower() in ("fourier", "positional")
    assert m.progressive.lower() == "none" and m.progressive_input.lower() in ("none", "residual")
    assert m.fir or m.progressive_input.lower() == "none"  # the pyramid downsample is restated for fir=True only
    idx = {"res": 0, "attn": 0, "down": 0}

    def RB(h, temb, **kw):
        p = params["ResnetBlockBigGANpp_%d" % idx["res"]]
        idx["res"] += 1
        return resnet_block_biggan(h, temb, p, cfg, **kw)

    def AT(h):
        p = params["AttnBlockpp_%d" % idx["attn"]]
        idx["attn"] += 1
        return attn_block(h, p, cfg)

    # :71-78 Fourier embedding of log(time_cond)
    if m.embedding_type.lower() == "fourier":
        Wf = _t(params["GaussianFourierProjection_0"]["W"], x)
        x_proj = torch.log(time_cond.to(x.dtype))[:, None] * Wf[None, :] * 2 * math.pi
        temb = torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)
    else:  # :80-84 sinusoidal positional embedding of the (continuous) labels
        temb = get_timestep_embedding(time_cond.to(x.dtype), nf)
    if m.conditional:  # :87-91
        temb = dense(temb, params["Dense_0"])
        temb = dense(swish(temb), params["Dense_1"])
    else:
        temb = None
    if not cfg.data.centered:
        x = 2 * x - 1.0  # :136-138
    input_pyramid = x if m.progressive_input.lower() != "none" else None
    hs = [conv_same(x, params["Conv_0"])]  # :146
    for i_level in range(nres):
        for _ in range(nrb):
            h = RB(hs[-1], temb, out_ch=nf * ch_mult[i_level])
            if h.shape[1] in attn_res:
                h = AT(h)
            hs.append(h)
        if i_level != nres - 1:
            h = RB(hs[-1], temb, down=True)
            if m.progressive_input.lower() == "residual":
                p = params["Downsample_%d" % idx["down"]]
                idx["down"] += 1
                input_pyramid = pyramid_downsample(input_pyramid, p, cfg)
                if m.skip_rescale:
                    input_pyramid = (input_pyramid + h) / math.sqrt(2.0)
                else:
                    input_pyramid = input_pyramid + h
                h = input_pyramid
            hs.append(h)
    h = hs[-1]
    h = RB(h, temb)
    h = AT(h)
    h = RB(h, temb)
    for i_level in reversed(range(nres)):
        for _ in range(nrb + 1):
            h = RB(torch.cat([h, hs.pop()], dim=-1), temb, out_ch=nf * ch_mult[i_level])
        if h.shape[1] in attn_res:
            h = AT(h)
        if i_level != 0:
            h = RB(h, temb, up=True)
    assert not hs
    h = swish(group_norm(h, params["GroupNorm_0"]))
    h = conv_same(h, params["Conv_1"])
    if m.scale_by_sigma:
        h = h / time_cond.to(x.dtype).reshape(-1, 1, 1, 1)
    return h


# ---- eps wrappers ----
def cld_eps_fn(params, cfg, dtype=torch.float32):
    """cld_jax/models/utils.py:128-182 with mixed_score=False: u (B,H,W,C,2) numpy -> eps."""

    def eps_fn(u, t):
        ut = torch.as_tensor(np.asarray(u), dtype=dtype)
        B, H, W, C, _ = ut.shape
        x6 = ut.permute(0, 1, 2, 4, 3).reshape(B, H, W, 2 * C)  # 'b ... d g -> b ... (g d)'
        labels = torch.as_tensor(np.asarray(t), dtype=dtype) * 999  # :172
        with torch.no_grad():
            out = ncsnpp_forward(params, x6, labels, cfg)
        out = out.reshape(B, H, W, 2, C).permute(0, 1, 2, 4, 3)  # 'b ... (g d) -> b ... d g'
        return out.numpy().astype(np.asarray(u).dtype)

    return eps_fn


def blur_eps_fn(params, cfg, dtype=torch.float32):
    """blur_jax/models/utils.py:141-153: labels = 999 t, encode_x / model2eps identities."""

    def eps_fn(x, t):
        xt = torch.as_tensor(np.asarray(x), dtype=dtype)
        labels = torch.as_tensor(np.asarray(t), dtype=dtype) * 999
        with torch.no_grad():
            out = ncsnpp_forward(params, xt, labels, cfg)
        return out.numpy().astype(np.asarray(x).dtype)

    return eps_fn


# ---- synthetic parameters with flax's tree structure ----
def param_shapes(cfg, in_ch):
    """Walk ncsnpp.py:41-243 and return {'A/B/C': shape} for every parameter, flax naming."""
    m = cfg.model
    nf, ch_mult, nrb = m.nf, tuple(m.ch_mult), m.num_res_blocks
    attn_res = tuple(m.attn_resolutions)
    nres = len(ch_mult)
    res = cfg.data.image_size
    shapes = {}
    cnt = {"res": 0, "attn": 0, "down": 0}

    def add_rb(cin, cout, resample):
        n = "ResnetBlockBigGANpp_%d" % cnt["res"]
        cnt["res"] += 1
        shapes[n + "/GroupNorm_0/scale"] = (cin,)
        shapes[n + "/GroupNorm_0/bias"] = (cin,)
        shapes[n + "/Conv_0/kernel"] = (3, 3, cin, cout)
        shapes[n + "/Conv_0/bias"] = (cout,)
        shapes[n + "/Dense_0/kernel"] = (nf * 4, cout)
        shapes[n + "/Dense_0/bias"] = (cout,)
        shapes[n + "/GroupNorm_1/scale"] = (cout,)
        shapes[n + "/GroupNorm_1/bias"] = (cout,)
        shapes[n + "/Conv_1/kernel"] = (3, 3, cout, cout)
        shapes[n + "/Conv_1/bias"] = (cout,)
        if cin != cout or resample:
            shapes[n + "/Conv_2/kernel"] = (1, 1, cin, cout)
            shapes[n + "/Conv_2/bias"] = (cout,)

    def add_attn(c):
        n = "AttnBlockpp_%d" % cnt["attn"]
        cnt["attn"] += 1
        shapes[n + "/GroupNorm_0/scale"] = (c,)
        shapes[n + "/GroupNorm_0/bias"] = (c,)
        for i in range(4):
            shapes[n + "/NIN_%d/W" % i] = (c, c)
            shapes[n + "/NIN_%d/b" % i] = (c,)

    if m.embedding_type.lower() == "fourier":
        shapes["GaussianFourierProjection_0/W"] = (nf,)
        shapes["Dense_0/kernel"] = (2 * nf, 4 * nf)
    else:  # positional: no parameters in the embedding, nf features
        shapes["Dense_0/kernel"] = (nf, 4 * nf)
    shapes["Dense_0/bias"] = (4 * nf,)
    shapes["Dense_1/kernel"] = (4 * nf, 4 * nf)
    shapes["Dense_1/bias"] = (4 * nf,)
    shapes["Conv_0/kernel"] = (3, 3, in_ch, nf)
    shapes["Conv_0/bias"] = (nf,)
    hs_c = [nf]
    c = nf
    pyr_c = in_ch
    cur_res = res
    for i_level in range(nres):
        for _ in range(nrb):
            add_rb(c, nf * ch_mult[i_level], False)
            c = nf * ch_mult[i_level]
            if cur_res in attn_res:
                add_attn(c)
            hs_c.append(c)
        if i_level != nres - 1:
            add_rb(c, c, True)
            cur_res //= 2
            if m.progressive_input.lower() == "residual":
                n = "Downsample_%d" % cnt["down"]
                cnt["down"] += 1
                shapes[n + "/Conv2d_0/weight"] = (3, 3, pyr_c, c)
                shapes[n + "/Conv2d_0/bias"] = (c,)
                pyr_c = c
            hs_c.append(c)
    add_rb(c, c, False)
    add_attn(c)
    add_rb(c, c, False)
    for i_level in reversed(range(nres)):
        for _ in range(nrb + 1):
            add_rb(c + hs_c.pop(), nf * ch_mult[i_level], False)
            c = nf * ch_mult[i_level]
        if cur_res in attn_res:
            add_attn(c)
        if i_level != 0:
            add_rb(c, c, True)
            cur_res *= 2
    assert not hs_c
    shapes["GroupNorm_0/scale"] = (c,)
    shapes["GroupNorm_0/bias"] = (c,)
    shapes["Conv_1/kernel"] = (3, 3, c, in_ch)
    shapes["Conv_1/bias"] = (in_ch,)
    return shapes


def random_params(cfg, in_ch, seed=0, dtype=np.float32):
    """Non-degenerate synthetic weights: fan-avg uniform kernels with init_scale 1.0 everywhere
    (flax's init_scale=0 -> 1e-10 would make eps ~ 0), random biases and GroupNorm
    affine so every term of the forward is exercised.  Returns a nested dict (flax tree)."""
    rng = np.random.default_rng(seed)
    flat = {}
    for name, shp in param_shapes(cfg, in_ch).items():
        leaf = name.split("/")[-1]
        if leaf in ("kernel", "weight", "W") and len(shp) >= 2:
            rf = int(np.prod(shp[:-2])) if len(shp) > 2 else 1
            fan_in, fan_out = shp[-2] * rf, shp[-1] * rf
            lim = math.sqrt(3.0 / ((fan_in + fan_out) / 2.0))
            a = rng.uniform(-lim, lim, size=shp)
        elif name == "GaussianFourierProjection_0/W":
            a = rng.standard_normal(shp) * cfg.model.fourier_scale
        elif leaf == "scale":
            a = 1.0 + 0.1 * rng.standard_normal(shp)
        else:  # biases
            a = 0.1 * rng.standard_normal(shp)
        flat[name] = a.astype(dtype)
    return unflatten(flat)


def unflatten(flat):
    tree = {}
    for name, a in flat.items():
        parts = name.split("/")
        d = tree
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = a
    return tree


def flatten(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        name = prefix + k
        if isinstance(v, dict):
            out.update(flatten(v, name + "/"))
        else:
            out[name] = v
    return out
