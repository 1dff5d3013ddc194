"""fp32 error of the Winograd variants of a 3x3 convolution, emulated on the CPU (numpy float32, sequential fp32 accumulation over the
channels -- the order of an MFMA chain), against a float64 direct convolution.  VERDICT r2 item 2: measure F(4x4,3x3) and the hybrid
F(4,3)_x x F(2,3)_y instead of arguing about them.

The activation-side transforms (B^T d B and A^T M A) run in float32 exactly as a kernel would issue them (one rounding per add / fma is
approximated by one rounding per numpy operation); the weight transform G g G^T is computed in float64 and rounded once, as
`wino2d_weights_kernel` does.  Output: a table per layer shape (Cin) of max|err| / max|out| and rms(err) / rms(out).

    python tools/wino_error_study.py [--out profiles/r03_winograd_error_study.md]
"""
import argparse
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from winograd_matrices import POINT_SETS, matrices, as_float  # noqa: E402
from fractions import Fraction as Fr  # noqa: E402


def f32(x):
    return np.asarray(x, dtype=np.float32)


def mat(name, scale=None):
    m, r, pts = POINT_SETS[name]
    AT, G, BT = matrices(m, r, pts, scale)
    return m, np.array(as_float(AT)), np.array(as_float(G)), np.array(as_float(BT))


def apply_f32(M, x, axis):
    """y = M @ x along `axis`, every partial sum rounded to float32 (left to right over the non-zero entries)."""
    x = np.moveaxis(x, axis, 0)
    out = []
    for row in M:
        acc = None
        for c, xs in zip(row, x):
            if c == 0:
                continue
            term = f32(np.float32(c) * xs)
            acc = term if acc is None else f32(acc + term)
        out.append(acc if acc is not None else np.zeros_like(x[0]))
    return np.moveaxis(np.stack(out), 0, axis)


def conv_direct64(x, w):
    B, H, W, C = x.shape
    xp = np.zeros((B, H + 2, W + 2, C))
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((B, H, W, w.shape[-1]))
    for ky in range(3):
        for kx in range(3):
            out += np.einsum("bhwc,cn->bhwn", xp[:, ky:ky + H, kx:kx + W], w[ky, kx])
    return out


def conv_direct32(x, w):
    """Direct fp32: one sequential accumulation over (tap, channel), as the implicit-GEMM kernel's MFMA chain."""
    B, H, W, C = x.shape
    xp = np.zeros((B, H + 2, W + 2, C), np.float32)
    xp[:, 1:-1, 1:-1] = x
    acc = np.zeros((B, H, W, w.shape[-1]), np.float32)
    w = f32(w)
    for ky in range(3):
        for kx in range(3):
            a = xp[:, ky:ky + H, kx:kx + W]
            for c in range(C):
                acc = f32(acc + f32(a[..., c, None] * w[ky, kx, c][None, None, None, :]))
    return acc


def conv_wino32(x, w, ny, nx):
    """Winograd with transform `ny` along y and `nx` along x (names of POINT_SETS, optional scale)."""
    my, ATy, Gy, BTy = mat(*ny)
    mx, ATx, Gx, BTx = mat(*nx)
    B, H, W, C = x.shape
    N = w.shape[-1]
    assert H % my == 0 and W % mx == 0
    py, px = my + 2, mx + 2
    xp = np.zeros((B, H + 2, W + 2, C), np.float32)
    xp[:, 1:-1, 1:-1] = x
    ty, tx = H // my, W // mx
    # patches d[b, ty, tx, r, s, c]
    d = np.zeros((B, ty, tx, py, px, C), np.float32)
    for r in range(py):
        for s in range(px):
            d[:, :, :, r, s] = xp[:, r:r + H:my, s:s + W:mx][:, :ty, :tx]
    V = apply_f32(BTx, d, 4)        # x direction first (in-lane in the kernels), then y
    V = apply_f32(BTy, V, 3)
    U = f32(np.einsum("ik,klcn,jl->ijcn", Gy, w.astype(np.float64), Gx))   # weights: double, one rounding
    M = np.zeros((B, ty, tx, py, px, N), np.float32)
    for c in range(C):               # sequential fp32 accumulation over the channels
        M = f32(M + f32(V[..., c, None] * U[None, None, None, :, :, c, :]))
    Y = apply_f32(ATx, M, 4)
    Y = apply_f32(ATy, Y, 3)         # (b, ty, tx, my, mx, n)
    return Y.transpose(0, 1, 3, 2, 4, 5).reshape(B, H, W, N)


VARIANTS = [
    ("direct fp32 (implicit GEMM)", None),
    ("F(2x2,3x3)  [production r2]", (("F23",), ("F23",))),
    ("hybrid F(2,3)y x F(4,3)x, points 0,+-1,+-2", (("F23",), ("F43_std",))),
    ("hybrid F(2,3)y x F(4,3)x, points 0,+-1,+-1/2", (("F23",), ("F43_half",))),
    ("hybrid F(2,3)y x F(4,3)x, points 0,+-1,1/2,-2", (("F23",), ("F43_mixed",))),
    ("F(4x4,3x3), points 0,+-1,+-2", (("F43_std",), ("F43_std",))),
    ("F(4x4,3x3), points 0,+-1,+-1/2", (("F43_half",), ("F43_half",))),
    ("F(4x4,3x3), points 0,+-1,1/2,-2", (("F43_mixed",), ("F43_mixed",))),
]


def study(cins, n_out, seed):
    rng = np.random.default_rng(seed)
    rows = []
    for cin in cins:
        x = rng.standard_normal((2, 16, 16, cin))
        x = f32(x / (1.0 + np.exp(-x)))                      # swish(GroupNorm output): what the 3x3 convolutions read
        w = rng.standard_normal((3, 3, cin, n_out)) / np.sqrt(9.0 * cin) * np.sqrt(2.0)
        w = f32(w).astype(np.float64)
        want = conv_direct64(x.astype(np.float64), w)
        sc, rms = np.abs(want).max(), np.sqrt((want ** 2).mean())
        for name, v in VARIANTS:
            got = conv_direct32(x, w) if v is None else conv_wino32(x, w, v[0], v[1])
            e = got.astype(np.float64) - want
            rows.append((cin, name, np.abs(e).max() / sc, np.sqrt((e ** 2).mean()) / rms))
            print("Cin=%4d  %-52s max|e|/max|y| = %.2e   rms(e)/rms(y) = %.2e" % rows[-1], flush=True)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cins", default="128,256,512")
    ap.add_argument("--nout", type=int, default=32)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rows = study([int(c) for c in a.cins.split(",")], a.nout, a.seed)
    if a.out:
        with open(a.out, "w") as f:
            f.write("# fp32 error of the Winograd variants of the 3x3 convolution (CPU emulation, tools/wino_error_study.py)\n\n")
            f.write("2 images of 16x16, input = swish(N(0,1)), weights N(0, 2/(9 Cin)), %d outputs; sequential fp32 accumulation over the\n"
                    "channels; error against a float64 direct convolution.\n\n" % a.nout)
            f.write("| Cin | algorithm | max err / max out | rms err / rms out | MFMA products per output |\n|---|---|---|---|---|\n")
            prod = {"direct": "9", "F(2x2": "4", "hybrid": "3", "F(4x4": "2.25"}
            for cin, name, emax, erms in rows:
                k = [v for p, v in prod.items() if name.startswith(p)][0]
                f.write("| %d | %s | %.2e | %.2e | %s |\n" % (cin, name, emax, erms, k))


if __name__ == "__main__":
    main()
