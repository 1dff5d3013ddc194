"""fp32 operands on the bf16 matrix core: the 3-piece split (hi / mid / lo bf16, 6 products, fp32 accumulation) that VERDICT r4 item 6 asks
to probe, emulated on the CPU next to the fp32-MFMA forms it would have to beat -- direct and Winograd-domain, error against float64.

    a = a_hi + a_mid + a_lo (each bf16, round-to-nearest-even of the running remainder: together 24 significand bits),
    a b ~= a_hi b_hi + (a_hi b_mid + a_mid b_hi) + (a_hi b_lo + a_lo b_hi + a_mid b_mid)      -- every dropped term <= 2^-24 |a b|
    bf16x3 (the engine's opt-in mode): hi / lo only, a_hi b_hi + a_hi b_lo + a_lo b_hi         -- dropped terms ~ 2^-16 |a b|

A bf16 x bf16 product is exact in fp32 (8 x 8 significand bits); the matrix core accumulates in fp32.  The emulation accumulates the
six products of one k step smallest first, then over k sequentially (an MFMA chain), every partial sum rounded to float32.

    python tools/split6_error_study.py [--out profiles/r05_split6_error_study.md]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from wino_error_study import apply_f32, conv_direct32, conv_direct64, f32, mat  # noqa: E402


def bf16(x):
    """Round float32 to bfloat16 (nearest even), returned as float32."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split(x, pieces):
    out, rem = [], f32(x)
    for _ in range(pieces):
        p = bf16(rem)
        out.append(p)
        rem = f32(rem - p)
    return out


TERMS = {2: [(1, 0), (0, 1), (0, 0)],                                   # bf16x3: lo*hi, hi*lo, hi*hi (smallest first)
         3: [(1, 1), (2, 0), (0, 2), (1, 0), (0, 1), (0, 0)]}           # split6: mid*mid, lo*hi, hi*lo, mid*hi, hi*mid, hi*hi


def contract_split(A, Bm, pieces):
    """sum_c A[..., c, None] * Bm[..., c, :] with both operands split into `pieces` bf16 pieces; fp32 accumulation over (c, term)."""
    As, Bs = split(A, pieces), split(Bm, pieces)
    acc = np.zeros(A.shape[:-1] + (Bm.shape[-1],), np.float32)
    for c in range(A.shape[-1]):
        for ia, ib in TERMS[pieces]:
            acc = f32(acc + f32(As[ia][..., c, None] * Bs[ib][..., c, :]))
    return acc


def conv_direct_split(x, w, pieces):
    B, H, W, C = x.shape
    xp = np.zeros((B, H + 2, W + 2, C), np.float32)
    xp[:, 1:-1, 1:-1] = x
    acc = np.zeros((B, H, W, w.shape[-1]), np.float32)
    xs, ws = split(xp, pieces), split(f32(w), pieces)
    for ky in range(3):
        for kx in range(3):
            for c in range(C):
                for ia, ib in TERMS[pieces]:
                    acc = f32(acc + f32(xs[ia][:, ky:ky + H, kx:kx + W, c, None] * ws[ib][ky, kx, c][None, None, None, :]))
    return acc


def conv_wino_split(x, w, ny, nx, pieces):
    """Winograd with the component GEMMs on split operands: V (fp32 transform of the activations, then split), U (float64 transform of
    the weights, rounded to fp32, then split)."""
    my, ATy, Gy, BTy = mat(*ny)
    mx, ATx, Gx, BTx = mat(*nx)
    B, H, W, C = x.shape
    N = w.shape[-1]
    py, px = my + 2, mx + 2
    xp = np.zeros((B, H + 2, W + 2, C), np.float32)
    xp[:, 1:-1, 1:-1] = x
    ty, tx = H // my, W // mx
    d = np.zeros((B, ty, tx, py, px, C), np.float32)
    for r in range(py):
        for s in range(px):
            d[:, :, :, r, s] = xp[:, r:r + H:my, s:s + W:mx][:, :ty, :tx]
    V = apply_f32(BTy, apply_f32(BTx, d, 4), 3)
    U = f32(np.einsum("ik,klcn,jl->ijcn", Gy, w.astype(np.float64), Gx))
    M = contract_split(V, np.broadcast_to(U[None, None, None], (B, ty, tx) + U.shape), pieces)
    Y = apply_f32(ATy, apply_f32(ATx, M, 4), 3)
    return Y.transpose(0, 1, 3, 2, 4, 5).reshape(B, H, W, N)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cins", default="128,256,512")
    ap.add_argument("--nout", type=int, default=16)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    F23, F43 = ("F23",), ("F43_std",)
    forms = [
        ("direct, fp32 MFMA (the bar of criterion 1)", 9.0, lambda x, w: conv_direct32(x, w)),
        ("hybrid F(2,3)y x F(4,3)x, fp32 MFMA (production)", 3.0, lambda x, w: __import__("wino_error_study").conv_wino32(x, w, F23, F43)),
        ("direct, bf16x3 (2 pieces, 3 products: the opt-in mode)", 27 / 16, lambda x, w: conv_direct_split(x, w, 2)),
        ("direct, split6 (3 pieces, 6 products)", 54 / 16, lambda x, w: conv_direct_split(x, w, 3)),
        ("F(2x2,3x3) in split6", 24 / 16, lambda x, w: conv_wino_split(x, w, F23, F23, 3)),
        ("hybrid F(2,3)y x F(4,3)x in split6", 18 / 16, lambda x, w: conv_wino_split(x, w, F23, F43, 3)),
    ]
    rows = []
    for cin in [int(c) for c in a.cins.split(",")]:
        x = rng.standard_normal((1, 8, 8, cin))
        x = f32(x / (1.0 + np.exp(-x)))
        w = f32(rng.standard_normal((3, 3, cin, a.nout)) / np.sqrt(9.0 * cin) * np.sqrt(2.0)).astype(np.float64)
        want = conv_direct64(x.astype(np.float64), w)
        sc, rms = np.abs(want).max(), np.sqrt((want ** 2).mean())
        for name, cost, fn in forms:
            e = fn(x, w).astype(np.float64) - want
            rows.append((cin, name, np.abs(e).max() / sc, np.sqrt((e ** 2).mean()) / rms, cost))
            print("Cin=%4d  %-58s max|e|/max|y| = %.2e  rms = %.2e  matrix time per output = %.2f fp32-MFMA products" % rows[-1], flush=True)
    if a.out:
        with open(a.out, "w") as f:
            f.write("# fp32 operands split into three bf16 pieces (6 products) against the fp32-MFMA forms: error (CPU emulation, tools/split6_error_study.py)\n\n"
                    "1 image of 8x8, input = swish(N(0,1)), weights N(0, 2/(9 Cin)), %d outputs; fp32 accumulation over (channel, term); error against a float64\n"
                    "direct convolution.  Matrix time per output pixel and output channel in units of one fp32-MFMA product (a bf16 product costs 1/16).\n\n" % a.nout)
            f.write("| Cin | form | max err / max out | rms err / rms out | matrix time per output |\n|---|---|---|---|---|\n")
            for r in rows:
                f.write("| %d | %s | %.2e | %.2e | %.2f |\n" % r)


if __name__ == "__main__":
    main()
