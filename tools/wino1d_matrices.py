"""Exact Toom-Cook matrices of the 1-D Winograd algorithms F(2,R) used along image rows for the 5x5 / 7x7 layers,
and an fp32 error estimate against an fp64 direct correlation.   y = AT [ (G g) (.) (BT d) ],  y_i = sum_j g_j d_{i+j}."""
from fractions import Fraction as Fr
import numpy as np


def poly_mul(a, b):
    out = [Fr(0)] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] += x * y
    return out


def matrices(m, r, pts):
    n = m + r - 1
    assert len(pts) == n - 1
    pts = [Fr(p) for p in pts]
    f = [np.prod([pts[k] - pts[l] for l in range(n - 1) if l != k]) for k in range(n - 1)]
    AT = [[pts[k] ** i for k in range(n - 1)] + [Fr(1 if i == m - 1 else 0)] for i in range(m)]
    G = [[pts[k] ** j / f[k] for j in range(r)] for k in range(n - 1)] + [[Fr(0)] * (r - 1) + [Fr(1)]]
    BT = []
    for k in range(n - 1):
        p = [Fr(1)]
        for l in range(n - 1):
            if l != k:
                p = poly_mul(p, [-pts[l], Fr(1)])
        BT.append(p + [Fr(0)] * (n - len(p)))
    p = [Fr(1)]
    for l in range(n - 1):
        p = poly_mul(p, [-pts[l], Fr(1)])
    BT.append(p)
    return AT, G, BT


def check(m, r, pts, trials=2000, seed=0):
    AT, G, BT = matrices(m, r, pts)
    n = m + r - 1
    rng = np.random.default_rng(seed)
    # exactness in rationals
    g = [Fr(int(x)) for x in rng.integers(-5, 6, r)]
    d = [Fr(int(x)) for x in rng.integers(-5, 6, n)]
    U = [sum(G[k][j] * g[j] for j in range(r)) for k in range(n)]
    V = [sum(BT[k][j] * d[j] for j in range(n)) for k in range(n)]
    y = [sum(AT[i][k] * U[k] * V[k] for k in range(n)) for i in range(m)]
    want = [sum(g[j] * d[i + j] for j in range(r)) for i in range(m)]
    assert y == want, (y, want)
    A64, G64, B64 = [np.array([[float(x) for x in row] for row in M]) for M in (AT, G, BT)]
    # fp32 emulation: K = 68*7 channel-rows accumulate in fp32 per frequency point
    K = 476
    errs_w, errs_d, scale = [], [], []
    for _ in range(trials // 50):
        gk = rng.standard_normal((K, r)) * (2.0 / (K * r)) ** 0.5
        dk = np.abs(rng.standard_normal((K, n)))                        # post-ReLU-like activations
        want = np.array([sum((gk[:, j] * dk[:, i + j]).sum() for j in range(r)) for i in range(m)])
        U32 = (gk @ G64.T).astype(np.float32)                            # packed once in fp64, stored fp32
        V32 = (dk.astype(np.float32) @ B64.T.astype(np.float32)).astype(np.float32)
        M32 = (U32 * V32).astype(np.float32).sum(0, dtype=np.float32)
        y32 = (A64.astype(np.float32) @ M32).astype(np.float32)
        d32 = np.array([np.float32(sum((gk[:, j].astype(np.float32) * dk[:, i + j].astype(np.float32)).astype(np.float32).sum(dtype=np.float32) for j in range(r))) for i in range(m)])
        errs_w.append(np.abs(y32 - want).max()); errs_d.append(np.abs(d32 - want).max()); scale.append(np.abs(want).max())
    print("F(%d,%d) pts %s: winograd fp32 err %.2e, direct fp32 err %.2e (output scale %.2f); max|BT| %.1f max|G| %.3f max|AT| %.0f" %
          (m, r, pts, np.max(errs_w), np.max(errs_d), np.mean(scale), np.abs(B64).max(), np.abs(G64).max(), np.abs(A64).max()))
    return AT, G, BT


if __name__ == "__main__":
    check(2, 3, [0, 1, -1])
    check(2, 5, [0, 1, -1, 2, -2])
    check(2, 5, [0, 1, -1, Fr(1, 2), -Fr(1, 2)])
    check(2, 5, [0, 1, -1, 2, -Fr(1, 2)])
    check(2, 7, [0, 1, -1, 2, -2, Fr(1, 2), -Fr(1, 2)])
    AT, G, BT = matrices(2, 7, [0, 1, -1, 2, -2, Fr(1, 2), -Fr(1, 2)])
    for name, M in (("AT", AT), ("G", G), ("BT", BT)):
        print(name); [print("  ", [str(x) for x in row]) for row in M]


def c_tables():
    """Print the constexpr tables pasted into cnmnet_amd/csrc/conv_winograd_rows.hip."""
    for r, pts in ((5, [0, 1, -1, 2, -2]), (7, [0, 1, -1, 2, -2, Fr(1, 2), -Fr(1, 2)])):
        AT, G, BT = matrices(2, r, pts)
        n = r + 1
        f = lambda x: ("%d" % x if x.denominator == 1 else "%d. / %d" % (x.numerator, x.denominator))
        print("// F(2,%d), interpolation points %s, inf" % (r, [str(p) for p in pts]))
        print("template <> struct RowWino<%d> {" % r)
        print("    static constexpr float BT[%d][%d] = {%s};" % (n, n, ", ".join("{" + ", ".join(f(x) for x in row) + "}" for row in BT)))
        print("    static constexpr float AT1[%d] = {%s};   // AT0 = {1, ..., 1, 0}" % (n, ", ".join(f(x) for x in AT[1])))
        print("    static constexpr double G[%d][%d] = {%s};" % (n, r, ", ".join("{" + ", ".join(f(x) for x in row) + "}" for row in G)))
        print("};")


if __name__ == "__main__":
    c_tables()
