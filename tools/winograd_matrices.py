"""Winograd / Toom-Cook transform matrices F(m, r) for a chosen set of interpolation points, in exact rational arithmetic.

For a 1-D correlation  y[t] = sum_k d[t + k] g[k]  (t < m, k < r, n = m + r - 1 inputs):
    y = A^T [ (G g) * (B^T d) ]
with n - 1 finite points p_0 .. p_{n-2} and the point at infinity:
    A^T[t][k] = p_k^t            (k < n-1),   A^T[t][n-1] = [t == m-1]
    G[k][j]   = p_k^j / N_k      (k < n-1),   G[n-1][j]   = [j == r-1],      N_k = prod_{l != k} (p_k - p_l)
    B^T[k][:] = coefficients of  prod_{l != k} (x - p_l)   (k < n-1),   B^T[n-1][:] = coefficients of prod_l (x - p_l)
(the transposed form of Toom-Cook polynomial multiplication: Lavin & Gray 2016; Barabasz et al. 2020 for the choice of points).
The diagonal freedom  G -> D G, B^T -> D^-1 B^T  is exposed as `scale`: it moves constants between the weight side (transformed once, in
double) and the activation side (transformed in fp32 on the GPU), which matters for the size of the fp32 rounding error.

`matrices(m, r, points, scale)` returns (AT, G, BT) as lists of Fractions; `check()` verifies the identity exactly.
tools/wino_error_study.py measures the fp32 error of the resulting algorithms; the kernels' constants come from here.
"""
from fractions import Fraction as Fr


def _poly_mul(a, b):
    out = [Fr(0)] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] += x * y
    return out


def matrices(m, r, points, scale=None):
    n = m + r - 1
    pts = [Fr(p) for p in points]
    assert len(pts) == n - 1 and len(set(pts)) == n - 1
    scale = [Fr(1)] * n if scale is None else [Fr(s) for s in scale]
    AT = [[Fr(0)] * n for _ in range(m)]
    G = [[Fr(0)] * r for _ in range(n)]
    BT = [[Fr(0)] * n for _ in range(n)]
    for k, p in enumerate(pts):
        Nk = Fr(1)
        poly = [Fr(1)]
        for l, q in enumerate(pts):
            if l != k:
                Nk *= p - q
                poly = _poly_mul(poly, [-q, Fr(1)])
        for t in range(m):
            AT[t][k] = p ** t
        for j in range(r):
            G[k][j] = p ** j / Nk * scale[k]
        for s in range(n - 1):
            BT[k][s] = poly[s] / scale[k]
    full = [Fr(1)]
    for q in pts:
        full = _poly_mul(full, [-q, Fr(1)])
    AT[m - 1][n - 1] = Fr(1)
    G[n - 1][r - 1] = scale[n - 1]
    for s in range(n):
        BT[n - 1][s] = full[s] / scale[n - 1]
    return AT, G, BT


def check(m, r, AT, G, BT):
    """Exact check of y = A^T[(G g) * (B^T d)] on unit vectors (the identity is bilinear)."""
    n = m + r - 1
    for a in range(n):
        for b in range(r):
            d = [Fr(int(i == a)) for i in range(n)]
            g = [Fr(int(j == b)) for j in range(r)]
            U = [sum(G[k][j] * g[j] for j in range(r)) for k in range(n)]
            V = [sum(BT[k][s] * d[s] for s in range(n)) for k in range(n)]
            y = [sum(AT[t][k] * U[k] * V[k] for k in range(n)) for t in range(m)]
            want = [sum(d[t + k] * g[k] for k in range(r)) for t in range(m)]
            assert y == want, (a, b, y, want)
    return True


# The point sets measured by tools/wino_error_study.py.  F(2,3) is the classical one (what conv3x3_wino*_f32.hip use).
POINT_SETS = {
    "F23": (2, 3, [0, 1, -1]),
    "F43_std": (4, 3, [0, 1, -1, 2, -2]),                      # Lavin & Gray
    "F43_half": (4, 3, [0, 1, -1, Fr(1, 2), -Fr(1, 2)]),      # the reciprocal set: large constants move to A^T
    "F43_mixed": (4, 3, [0, 1, -1, Fr(1, 2), -2]),            # Barabasz et al.: one point < 1, one > 1
    "F43_mixed2": (4, 3, [0, 1, -1, 2, -Fr(1, 2)]),
}


def as_float(M):
    return [[float(x) for x in row] for row in M]


if __name__ == "__main__":
    for name, (m, r, pts) in POINT_SETS.items():
        AT, G, BT = matrices(m, r, pts)
        check(m, r, AT, G, BT)
        print(name, "points", [str(p) for p in pts])
        for lab, M in (("A^T", AT), ("G", G), ("B^T", BT)):
            print("  " + lab)
            for row in M:
                print("    [" + ", ".join("%7s" % str(x) for x in row) + "]")
