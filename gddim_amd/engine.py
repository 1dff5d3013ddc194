"""Python handle on a native `gd_engine` (include/gddim_amd.h).

PyTorch is used only as plumbing: device tensors (`.data_ptr()`), the current HIP stream and
`torch.distributed`.  All arithmetic on the sampling path runs in the HIP kernels behind the C ABI;
there is no CPU fallback -- without the native library or a GPU the compute calls raise GddimError.
"""
import ctypes
import json

import numpy as np

from . import _abi
from ._abi import GddimError, ModelDesc, check


def model_desc_from_config(config, in_channels, max_batch):
    """The architecture switches NCSNpp.__call__ reads (cld_jax/models/ncsnpp.py:41-69)."""
    m = config.model
    if m.name != 'ncsnpp':
        raise GddimError("only model.name == 'ncsnpp' is built (got %r)" % (m.name,))
    if m.resblock_type.lower() != 'biggan' or (m.fir and list(m.fir_kernel) != [1, 3, 3, 1]):
        raise GddimError("only resblock_type='biggan' (fir=False, or fir=True with fir_kernel=[1,3,3,1]) is built")
    if m.embedding_type.lower() not in ('fourier', 'positional') or m.progressive.lower() != 'none':
        raise GddimError("only embedding_type in {'fourier','positional'} with progressive='none' is built")
    if m.progressive_input.lower() not in ('none', 'residual') or m.nonlinearity.lower() != 'swish':
        raise GddimError("only progressive_input in {'none','residual'} and nonlinearity='swish' are built")
    if not m.resamp_with_conv:
        raise GddimError("only resamp_with_conv=True is built")
    d = ModelDesc()
    d.image_size = int(config.data.image_size)
    d.in_channels = int(in_channels)
    d.nf = int(m.nf)
    d.num_res_blocks = int(m.num_res_blocks)
    d.num_levels = len(m.ch_mult)
    for i, c in enumerate(m.ch_mult):
        d.ch_mult[i] = int(c)
    d.num_attn_resolutions = len(m.attn_resolutions)
    for i, r in enumerate(m.attn_resolutions):
        d.attn_resolutions[i] = int(r)
    d.conditional = int(bool(m.conditional))
    d.skip_rescale = int(bool(m.skip_rescale))
    d.progressive_input_residual = int(m.progressive_input.lower() == 'residual')
    d.scale_by_sigma = int(bool(m.scale_by_sigma))
    d.centered = int(bool(config.data.centered))
    d.max_batch = int(max_batch)
    d.positional_embedding = int(m.embedding_type.lower() == 'positional')
    d.no_fir = int(not m.fir)
    return d


def flatten_params(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        if hasattr(v, "items"):
            out.update(flatten_params(v, prefix + k + "/"))
        else:
            out[prefix + k] = v
    return out


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream(device):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _host_f32(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    return a, a.ctypes.data_as(ctypes.c_void_p)


class Engine(object):
    """One engine = one GPU, one architecture, one parameter set, batches up to max_batch."""

    def __init__(self, config, in_channels, max_batch, device=0, plan_only=False):
        self._lib = _abi.lib()
        self.config = config
        self.in_channels = int(in_channels)
        self.max_batch = int(max_batch)
        self.image_size = int(config.data.image_size)
        self.plan_only = bool(plan_only)
        if not plan_only:
            import torch
            if not torch.cuda.is_available() or self._lib.gd_device_count() == 0:
                raise GddimError("no MI355X / HIP device visible: the gddim_amd sampling engine has no CPU fallback")
            self.device = torch.device("cuda", device if isinstance(device, int) else device.index or 0)
            dev_index = self.device.index
        else:
            self.device = None
            dev_index = -1
        desc = model_desc_from_config(config, in_channels, max_batch)
        h = ctypes.c_void_p()
        check(self._lib.gd_engine_create(ctypes.byref(desc), dev_index, ctypes.byref(h)))
        self._h = h
        self._schedule_token = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gd_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- introspection ----
    def plan(self):
        need = ctypes.c_size_t()
        check(self._lib.gd_engine_plan_json(self._h, None, 0, ctypes.byref(need)))
        buf = ctypes.create_string_buffer(need.value)
        check(self._lib.gd_engine_plan_json(self._h, buf, need.value, None))
        return json.loads(buf.value.decode())

    def flops_per_image(self, conv3x3_only=False):
        return self._lib.gd_engine_flops_per_image(self._h, int(conv3x3_only))

    def param_specs(self):
        out = {}
        name = ctypes.create_string_buffer(256)
        shape = (ctypes.c_int64 * 4)()
        nd = ctypes.c_int()
        for i in range(self._lib.gd_engine_num_params(self._h)):
            check(self._lib.gd_engine_param_info(self._h, i, name, 256, shape, ctypes.byref(nd)))
            out[name.value.decode()] = tuple(int(shape[j]) for j in range(nd.value))
        return out

    def weight_bytes(self):
        """Device bytes of the weights: packed GEMM layout, F(2x2,3x3) copies, hybrid Winograd copies, bf16 hi + lo copies."""
        out = (ctypes.c_longlong * 4)()
        check(self._lib.gd_engine_weight_bytes(self._h, out))
        return {"packed": out[0], "wino2d": out[1], "wino42": out[2], "bf16_split": out[3]}

    def packed_weight(self, slot):
        n = ctypes.c_size_t()
        check(self._lib.gd_engine_packed_weight(self._h, slot.encode(), None, 0, ctypes.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        check(self._lib.gd_engine_packed_weight(self._h, slot.encode(), out.ctypes.data_as(ctypes.c_void_p), n.value, None))
        return out

    # ---- weights ----
    def load_params(self, params_tree):
        """params_tree: the flax `params_ema` tree (nested dict of arrays) or a flat {'a/b/c': array}."""
        flat = flatten_params(params_tree)
        specs = self.param_specs()
        extra = sorted(set(flat) - set(specs))
        if extra:
            raise GddimError("unexpected parameters (not part of this architecture): %s" % ", ".join(extra[:8]))
        for name, arr in flat.items():
            if hasattr(arr, "detach"):
                arr = arr.detach().cpu().numpy()
            a, p = _host_f32(arr)
            shp = (ctypes.c_int64 * max(1, a.ndim))(*a.shape)
            check(self._lib.gd_engine_load_param(self._h, name.encode(), p, shp, a.ndim))
        check(self._lib.gd_engine_params_ready(self._h))
        self._schedule_token = None
        return self

    def set_precision(self, mode):
        """'fp32' (fp32 MFMA, 3x3 convolutions through the hybrid Winograd transform F(2,3)y x F(4,3)x; default), 'bf16x3' (split-precision
        weight contractions on the bf16 matrix core) or 'split6' (fp32 mode with the 3x3 convolutions of the 16x16-tileable levels through
        F(2x2,3x3) in the exact three-piece bf16 split: fp32-class error, round 6)."""
        code = {"fp32": 0, "f32": 0, 0: 0, "bf16x3": 1, 1: 1, "split6": 2, 2: 2}[mode]
        check(self._lib.gd_engine_set_precision(self._h, code))
        self._schedule_token = None
        return self

    # ---- score network ----
    def unet(self, x, labels, out=None):
        """eps = NCSNpp(x, labels); x (B,H,W,in_channels) fp32 on this engine's GPU."""
        import torch
        x = x.contiguous()
        labels = labels.to(torch.float32).contiguous()
        B = x.shape[0]
        if out is None:
            out = torch.empty_like(x)
        check(self._lib.gd_unet_forward(self._h, _ptr(x), _ptr(labels), _ptr(out), B, _stream(self.device)))
        return out

    def cld_eps(self, u, t, out=None):
        """eps_fn(u, t) of cld_jax/models/utils.py:168-182: u (B,H,W,C,2), t (B,)."""
        import torch
        u = u.contiguous()
        t = t.to(torch.float32).contiguous()
        if out is None:
            out = torch.empty_like(u)
        check(self._lib.gd_cld_eps(self._h, _ptr(u), _ptr(t), _ptr(out), u.shape[0], _stream(self.device)))
        return out

    # ---- samplers ----
    def set_schedule_cld(self, ts, coef, denoise, t_denoise=0.0, dn_F=None, dn_GG=None, dn_RinvT=None, token=None):
        if token is not None and token == self._schedule_token:
            return
        coef = np.asarray(coef, dtype=np.float32)
        num_step, n_slots = coef.shape[0], coef.shape[1]
        ts_a, ts_p = _host_f32(ts)
        assert ts_a.shape[0] == num_step
        co_a, co_p = _host_f32(coef)
        keep = [_host_f32(m if m is not None else np.zeros(4)) for m in (dn_F, dn_GG, dn_RinvT)]
        check(self._lib.gd_engine_set_schedule_cld(self._h, num_step, n_slots, ts_p, co_p, int(bool(denoise)),
                                                   float(t_denoise), keep[0][1], keep[1][1], keep[2][1]))
        self._schedule_token = token

    def set_transforms_cld(self, pre=None, eps_lin=None, u_lin=None, dn_u_lin=None, post=None):
        """Per-step 2x2 maps of the installed CLD schedule (include/gddim_amd.h: gd_engine_set_transforms_cld)."""
        keep = [None if m is None else _host_f32(m) for m in (pre, eps_lin, u_lin, dn_u_lin, post)]
        check(self._lib.gd_engine_set_transforms_cld(self._h, *[None if k is None else k[1] for k in keep]))

    def set_noise_cld(self, pre_lin=None, pre_noise=None, post_noise=None):
        """Per-step noise maps of the stochastic CLD samplers (include/gddim_amd.h: gd_engine_set_noise_cld)."""
        keep = [None if m is None else _host_f32(m) for m in (pre_lin, pre_noise, post_noise)]
        check(self._lib.gd_engine_set_noise_cld(self._h, *[None if k is None else k[1] for k in keep]))

    def set_seed(self, seed):
        check(self._lib.gd_engine_set_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF))

    def sample_cld(self, u, inverse_scale=True, use_graph=True):
        """use_graph: False eager launches, True one captured graph per step replayed num_step times, 'loop' one graph for the
        whole call (include/gddim_amd.h: gd_sample_cld)."""
        import torch
        use_graph = 2 if use_graph == 'loop' else int(bool(use_graph))
        u = u.contiguous()
        B = u.shape[0]
        x = torch.empty(u.shape[:-1], dtype=torch.float32, device=u.device)
        v = torch.empty_like(x)
        check(self._lib.gd_sample_cld(self._h, B, _ptr(u), _ptr(x), _ptr(v), int(bool(inverse_scale)), use_graph,
                                      _stream(self.device)))
        return x, v

    def set_schedule_blur(self, ts, mean_coef, std_coef, token=None):
        if token is not None and token == self._schedule_token:
            return
        ts_a, ts_p = _host_f32(ts)
        m_a, m_p = _host_f32(mean_coef)
        s_a, s_p = _host_f32(std_coef)
        nfe = ts_a.shape[0]
        assert m_a.size == (nfe + 1) * 1024 and s_a.size == nfe + 1
        check(self._lib.gd_engine_set_schedule_blur(self._h, nfe, ts_p, m_p, s_p))
        self._schedule_token = token

    def set_schedule_blur_deis(self, ts, coef, token=None):
        """coef (nfe, n_slots, 32, 32): slot 0 the transition, slots 1.. the eps coefficients (newest first)."""
        if token is not None and token == self._schedule_token:
            return
        co_a, co_p = _host_f32(coef)
        ts_a, ts_p = _host_f32(ts)
        nfe, n_slots = co_a.shape[0], co_a.shape[1]
        assert ts_a.shape[0] == nfe and co_a.shape[2:] == (32, 32)
        check(self._lib.gd_engine_set_schedule_blur_deis(self._h, nfe, n_slots, ts_p, co_p))
        self._schedule_token = token

    def set_model2eps_blur(self, rows=None):
        """Per-step rows [a, b(32*32)] of sde.model2eps in linear form, eps_y = a * DCT(model_output) + b (.) y
        (include/gddim_amd.h); None removes it."""
        if rows is None:
            check(self._lib.gd_engine_set_model2eps_blur(self._h, None))
            return
        a, p = _host_f32(rows)
        assert a.ndim == 2 and a.shape[1] == 1025
        check(self._lib.gd_engine_set_model2eps_blur(self._h, p))

    def sample_blur(self, y, inverse_scale=True, use_graph=True):
        """use_graph as in sample_cld: False eager, True one graph per step replayed nfe times, 'loop' one graph for the whole call."""
        import torch
        use_graph = 2 if use_graph == 'loop' else int(bool(use_graph))
        y = y.contiguous()
        x = torch.empty_like(y)
        check(self._lib.gd_sample_blur(self._h, y.shape[0], _ptr(y), _ptr(x), int(bool(inverse_scale)), use_graph,
                                       _stream(self.device)))
        return x

    # ---- profiling ----
    def profile(self, enable):
        check(self._lib.gd_engine_profile(self._h, int(bool(enable))))

    def profile_read(self):
        n = ctypes.c_int()
        arr = (_abi.ProfEntry * 256)()
        check(self._lib.gd_engine_profile_read(self._h, arr, 256, ctypes.byref(n)))
        return [{"name": arr[i].name.decode(), "ms": arr[i].ms, "flops": arr[i].flops, "bytes": arr[i].bytes,
                 "launches": arr[i].launches} for i in range(min(n.value, 256))]
