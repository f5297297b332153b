#!/usr/bin/env python3
"""Generator of the polynomial constants in include/okenv_math.h (ok_sincosf, ok_tanhf), kept so that they can be re-derived.

For each kernel function the polynomial interpolates the target at Chebyshev nodes of the interval (near-minimax: within a small
factor of the best polynomial), in 200-digit arithmetic (mpmath); the coefficients are rounded to double and the error of the
ROUNDED polynomial, evaluated exactly, is reported relative to the function value the header forms from it.

  sin r = r + (r z) S(z),  S(z) = s1 + s2 z + ... + s5 z^4,        z = r^2, |r| <= 0.79 (pi/4 = 0.7854 plus the reduction's slack)
  cos r = 1 + z C(z),      C(z) = c1 + c2 z + ... + c6 z^5
  exp r - 1 = r + r^2 E(r), E(r) = e2 + e3 r + ... ,                |r| <= 0.3470 (ln 2 / 2 = 0.34657 plus slack)
"""
import sys

import mpmath as mp

mp.mp.dps = 200


def cheb_fit(f, a, b, n):
    """Coefficients (ascending powers) of the degree-(n-1) polynomial through f at the n Chebyshev nodes of [a, b]."""
    xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    A = mp.matrix(n, n)
    y = mp.matrix(n, 1)
    for i, x in enumerate(xs):
        for j in range(n):
            A[i, j] = x ** j
        y[i] = f(x)
    c = mp.lu_solve(A, y)
    return [c[i] for i in range(n)]


def to_double(c):
    return [float(v) for v in c]  # mpf -> nearest double


def horner(c, x):
    acc = mp.mpf(0)
    for v in reversed(c):
        acc = acc * x + mp.mpf(v)
    return acc


def max_rel_err(approx, exact, a, b, m=4001):
    worst = mp.mpf(0)
    for k in range(1, m):
        x = a + (b - a) * mp.mpf(k) / m
        e = abs(approx(x) - exact(x)) / abs(exact(x))
        worst = max(worst, e)
    return worst


def main():
    R = mp.mpf("0.79")
    # sin: S(z) = (sin(r)/r - 1) / z
    S = lambda z: (mp.sin(mp.sqrt(z)) / mp.sqrt(z) - 1) / z if z > 0 else mp.mpf(-1) / 6  # noqa: E731
    C = lambda z: (mp.cos(mp.sqrt(z)) - 1) / z if z > 0 else mp.mpf(-1) / 2  # noqa: E731
    for name, f, n in (("sin", S, int(sys.argv[1]) if len(sys.argv) > 1 else 5), ("cos", C, int(sys.argv[2]) if len(sys.argv) > 2 else 6)):
        c = to_double(cheb_fit(f, mp.mpf("1e-30"), R * R, n))
        if name == "sin":
            err = max_rel_err(lambda r: r + r * r * r * horner(c, r * r), mp.sin, mp.mpf("1e-6"), R)
        else:
            err = max_rel_err(lambda r: 1 + r * r * horner(c, r * r), mp.cos, mp.mpf("1e-6"), R)
        print("%s: %d coefficients, max relative error of the rounded polynomial on (0, %s]: %s = 2^%.1f" % (name, n, R, mp.nstr(err, 3), float(mp.log(err, 2))))
        for i, v in enumerate(c):
            print("    %s%d = %s  /* %s */" % (name[0], i + 1, v.hex(), repr(v)))
    L = mp.mpf("0.3470")
    E = lambda r: (mp.expm1(r) - r) / (r * r) if abs(r) > mp.mpf("1e-40") else mp.mpf(1) / 2 + r / 6  # noqa: E731
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 9
    c = to_double(cheb_fit(E, -L, L, n))
    err = max(max_rel_err(lambda r: r + r * r * horner(c, r), mp.expm1, mp.mpf("1e-6"), L), max_rel_err(lambda r: r + r * r * horner(c, r), mp.expm1, -L, mp.mpf("-1e-6")))
    print("expm1: %d coefficients, max relative error on [-%s, %s]: %s = 2^%.1f" % (n, L, L, mp.nstr(err, 3), float(mp.log(err, 2))))
    for i, v in enumerate(c):
        print("    e%d = %s  /* %s */" % (i + 2, v.hex(), repr(v)))
    # pi/2 = P1 + P2 (double-double), 2/pi; ln 2 = L1 + L2, 1/ln 2
    p1 = float(mp.pi / 2)
    p2 = float(mp.pi / 2 - mp.mpf(p1))
    print("pi/2: P1 = %s (%r), P2 = %s (%r); 2/pi = %r" % (p1.hex(), p1, p2.hex(), p2, float(2 / mp.pi)))
    l1 = float(mp.log(2))
    l2 = float(mp.log(2) - mp.mpf(l1))
    print("ln 2: L1 = %s (%r), L2 = %s (%r); 1/ln2 = %r" % (l1.hex(), l1, l2.hex(), l2, float(1 / mp.log(2))))


if __name__ == "__main__":
    main()
