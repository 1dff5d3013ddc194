"""The gate of VERDICT r5 item 2: gd_conv3x3_winograd2d_split6 against the production hybrid kernel on 16x16 256->256 and 32x32 128->128 at
B = 256 -- time per launch (hipEvents around `reps` back-to-back launches, weights pre-transformed for both, the two kernels alternating,
median of three) and rms error against float64
on 4 images.  python tools/bench_split6.py [reps]   (under rocprofv3 --kernel-trace --stats the per-kernel averages are the cross-check)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gddim_amd import _abi, ops  # noqa: E402
from gddim_amd.ops import _ptr, _stream  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cold = len(sys.argv) > 2 and "cold" in sys.argv[2:]  # rotate over 8 input / weight / output sets (> 1 GB: nothing stays in L2 or the 256 MB MALL)
with_res = "res" in sys.argv[2:]  # residual + 1/sqrt(2) output scale, as the second convolution of a residual block runs
dev = torch.device("cuda", 0)
lib = _abi.lib()
out = {}
for H, C in ((16, 256), (32, 128)):
    B, N = 256, C
    g = torch.Generator(device=dev).manual_seed(H)
    x = torch.randn((B, H, H, C), generator=g, device=dev)
    x = x * torch.sigmoid(x)
    w = torch.randn((N, 9 * C), generator=g, device=dev) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn((N,), generator=g, device=dev) * 0.1
    u6 = ops.conv3x3_winograd2d_split6_weights(w)
    y6 = torch.empty((B, H, H, N), device=dev)
    yh = torch.empty_like(y6)
    r = torch.randn((B, H, H, N), generator=g, device=dev) if with_res else None
    rp = _ptr(r) if with_res else None
    sc = 0.5 ** 0.5 if with_res else 1.0

    NS = 8 if cold else 1
    x_sets = [x] + [x.clone() for _ in range(NS - 1)]
    u6s = [u6] + [u6.clone() for _ in range(NS - 1)]
    y6s = [y6] + [torch.empty_like(y6) for _ in range(NS - 1)]
    cnt = [0, 0]

    def run6():
        i = cnt[0] % NS; cnt[0] += 1
        _abi.check(lib.gd_conv3x3_winograd2d_split6(_ptr(x_sets[i]), _ptr(u6s[i]), _ptr(b), rp, _ptr(y6s[i]), B, H, H, C, N, sc, _stream(x)))

    # the hybrid kernel with its weights transformed once, as the engine runs it (the stand-alone operator re-transforms them on every call)
    uh = torch.empty(24 * N * C, device=dev)
    _abi.check(lib.gd_conv3x3_winograd42_weights(_ptr(w), _ptr(uh), C, N, _stream(x)))

    uhs = [uh] + [uh.clone() for _ in range(NS - 1)]

    def runh():
        i = cnt[1] % NS; cnt[1] += 1
        _abi.check(lib.gd_conv3x3_winograd42_pretransformed(_ptr(x_sets[i]), _ptr(uhs[i]), _ptr(b), rp, _ptr(y6s[(i + 1) % NS] if cold else yh), B, H, H, C, N, 1.0, sc, _stream(x)))

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return a.elapsed_time(e) * 1e3 / n  # us

    # alternate the two kernels: a box's clock drifts with what ran before
    t6s, ths = [], []
    for _ in range(3):
        t6s.append(timed(run6, reps))
        ths.append(timed(runh, reps))
    t6, th = sorted(t6s)[1], sorted(ths)[1]
    # error on 4 images
    xs, want = x[:4].cpu(), None
    wd = w.cpu().double().reshape(N, 3, 3, C).permute(0, 3, 1, 2)
    want = torch.nn.functional.conv2d(xs.double().permute(0, 3, 1, 2), wd, bias=b.cpu().double(), padding=1).permute(0, 2, 3, 1)
    if with_res:
        want = (want + r[:4].cpu().double()) * sc
    cnt[0] = cnt[1] = 0
    if cold:
        _abi.check(lib.gd_conv3x3_winograd42_pretransformed(_ptr(x), _ptr(uh), _ptr(b), rp, _ptr(yh), B, H, H, C, N, 1.0, sc, _stream(x)))
    else:
        runh()
    run6()
    torch.cuda.synchronize()
    rms = lambda a: float(((a[:4].cpu().double() - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt())  # noqa: E731
    flops = 2.0 * B * H * H * C * N * 9
    key = "%dx%d_%d" % (H, H, C)
    out[key] = {"split6_us": t6, "hybrid_us": th, "speedup": th / t6, "split6_us_runs": t6s, "hybrid_us_runs": ths,
                "split6_algorithmic_tflops": flops / t6 / 1e6, "hybrid_algorithmic_tflops": flops / th / 1e6,
                "rms_rel_error_split6": rms(y6), "rms_rel_error_hybrid": rms(yh), "max_abs_diff_split6_vs_hybrid": float((y6 - yh).abs().max())}
    print(key, json.dumps(out[key]), flush=True)
print(json.dumps(out))
