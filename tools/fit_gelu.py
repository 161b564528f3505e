"""Derive and check the one-transcendental GELU of csrc/i2r_hrformer_lp.hip (gelu4).

GELU(x) = x Phi(x) = max(x, 0) - |x| Phi(-|x|), and the normal tail Phi(-a) = erfc(a / sqrt 2) / 2 is written 2^-Q(a) with Q a
degree-5 polynomial, Q(0) = 1 (so the tail is exactly 1/2 at 0).  The fit minimises the maximum error of the GELU VALUE,
a |2^-Q(a) - Phi(-a)|, over a in [0, 10]; the kernel evaluates Q by even / odd parts in w = x^2 (packed FMAs, |x| only once):
Q = (1 + c2 w + c4 w^2) + |x| (c1 + c3 w + c5 w^2), and GELU = x/2 + |x| (1/2 - 2^-Q).  The script prints the coefficients and the
maximum error of exactly that evaluation order in float32 against the float64 erf form (CPU only; numpy + scipy)."""
import numpy as np
from scipy.optimize import least_squares
from scipy.special import erf, erfc


def fit(deg=5):
    a = np.linspace(0, 10, 40001)
    true = 0.5 * erfc(a / np.sqrt(2))
    L = -np.log2(true)
    V = np.vander(a, deg + 1, increasing=True)[:, 1:]
    w = np.maximum(a * true * np.log(2), 1e-13)
    c, *_ = np.linalg.lstsq(V * w[:, None], (L - 1) * w, rcond=None)
    err = lambda c: a * (2.0 ** (-(1 + V @ c)) - true)
    for p in (4, 8, 16, 32):  # p-norms of growing order approach the minimax fit
        c = least_squares(lambda c: (np.abs(err(c)) * 1e6) ** (p / 2), c, method="lm", max_nfev=20000, xtol=1e-15, ftol=1e-15).x
    return c, np.abs(err(c)).max()


def gelu4_f32(x, c):
    """the kernel's evaluation order in float32"""
    f = np.float32
    c = c.astype(np.float32)
    x = x.astype(np.float32)
    w = np.minimum((x * x).astype(f), f(256))  # (clamped: the tail is 0 from |x| = 16 on, and x^2 cannot overflow)
    ev = (w * c[3] + c[1]).astype(f)
    ev = (ev * w + f(1)).astype(f)
    od = (w * c[4] + c[2]).astype(f)
    od = (od * w + c[0]).astype(f)
    a = np.abs(x)
    q = (a * od + ev).astype(f)
    t = (f(0.5) - np.exp2(-q).astype(f)).astype(f)
    return (a * t + f(0.5) * x).astype(f)


if __name__ == "__main__":
    c, e = fit()
    print("coefficients c1..c5:", ", ".join("%.8e" % v for v in c), " fit max |GELU error| %.2e" % e)
    x = np.concatenate([np.linspace(-12, 12, 4000001), [-1e4, 1e4, -50.0, 50.0, 0.0, 1e-30, -1e-30, 3e38, -3e38]])
    ref = 0.5 * x * (1 + erf(x / np.sqrt(2)))
    got = gelu4_f32(x, c).astype(np.float64)
    d = np.abs(got - ref)
    print("float32 evaluation: max |error| %.2e at x = %.4f; tail values %s" % (d[:-9].max(), x[d[:-9].argmax()], got[-9:]))
    x16 = np.linspace(-8, 8, 200001)
    print("max |error| / max(|GELU|, 1): %.2e" % (np.abs(gelu4_f32(x16, c) - 0.5 * x16 * (1 + erf(x16 / np.sqrt(2)))) / np.maximum(1, np.abs(x16))).max())
