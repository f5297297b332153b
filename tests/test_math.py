"""ok_sincosf (include/okenv_math.h): faithful to glibc sinf/cosf (what the reference's host code calls) and to
an fp64 reference; Philox known-answer test."""
import ctypes as C

import numpy as np

import _oracle as O


def ulp_diff(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def test_sincos_against_fp64_and_glibc(oracle):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-10, 10, 300000), rng.uniform(-2e4, 2e4, 300000), rng.uniform(-1.5e6, 1.5e6, 100000),
                        np.deg2rad(np.arange(-3600, 3601, 1, dtype=np.float64))]).astype(np.float32)
    s, c = np.zeros_like(x), np.zeros_like(x)
    O.lib().oracle_sincosf(x, s, c, x.size)
    sr = np.sin(x.astype(np.float64)).astype(np.float32)
    cr = np.cos(x.astype(np.float64)).astype(np.float32)
    # correctly rounded on this sample (the fp64 evaluation error is ~1e-16)
    assert ulp_diff(s, sr).max() <= 1 and (s != sr).mean() < 1e-5
    assert ulp_diff(c, cr).max() <= 1 and (c != cr).mean() < 1e-5
    libm = C.CDLL("libm.so.6")
    libm.sinf.restype = libm.cosf.restype = C.c_float
    libm.sinf.argtypes = libm.cosf.argtypes = [C.c_float]
    sub = x[::40]
    sg = np.array([libm.sinf(float(v)) for v in sub], dtype=np.float32)
    cg = np.array([libm.cosf(float(v)) for v in sub], dtype=np.float32)
    assert ulp_diff(s[::40], sg).max() <= 1
    assert ulp_diff(c[::40], cg).max() <= 1
    assert np.abs(s[::40].astype(np.float64) - sg).max() < 1e-7


def test_sincos_at_the_floats_nearest_to_multiples_of_half_pi(oracle):
    """Where the argument reduction is hardest: floats x = m * 2^e (m < 2^24) lying unusually close to a multiple of pi/2 -- the
    continued-fraction convergents of 2^e / (pi/2) give them for every exponent -- up to 2^31, the limit of the fast path.  The
    two-FMA reduction keeps a RELATIVE error of ~2^-52 on the reduced angle there, so the result is still the rounded fp64 one
    (compared here with a 60-digit evaluation, since numpy's own fp64 sine is not trustworthy for such arguments)."""
    import mpmath as mp
    mp.mp.dps = 60
    xs = set()
    half_pi = mp.pi / 2
    for e in range(-30, 8):
        alpha = mp.mpf(2) ** e / half_pi  # m * alpha close to an integer <=> m * 2^e close to a multiple of pi/2
        # convergents p/q of alpha with q < 2^24
        a, h0, h1, k0, k1 = alpha, 0, 1, 1, 0
        for _ in range(40):
            ai = int(mp.floor(a))
            h0, h1 = h1, ai * h1 + h0
            k0, k1 = k1, ai * k1 + k0
            if k1 >= (1 << 24):
                break
            if k1 > 0:
                x = float(k1) * 2.0 ** e
                if 0 < x < 2147483648.0:
                    xs.add(np.float32(x))
            frac = a - ai
            if frac == 0:
                break
            a = 1 / frac
    x = np.array(sorted(xs), dtype=np.float32)
    x = np.concatenate([x, -x])
    assert x.size > 400
    s, c = np.zeros_like(x), np.zeros_like(x)
    O.lib().oracle_sincosf(x, s, c, x.size)
    sr = np.array([float(mp.sin(mp.mpf(float(v)))) for v in x]).astype(np.float32)
    cr = np.array([float(mp.cos(mp.mpf(float(v)))) for v in x]).astype(np.float32)
    assert np.array_equal(s.view(np.uint32), sr.view(np.uint32)) and np.array_equal(c.view(np.uint32), cr.view(np.uint32))
    # how close they get: the smallest reduced angle in the set, and its sine still carries full precision
    rmin = min(abs(mp.mpf(float(v)) - mp.nint(mp.mpf(float(v)) / half_pi) * half_pi) for v in x if v > 0)
    assert rmin < 1e-8


def test_sincos_special_values(oracle):
    x = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 3.0e38], dtype=np.float32)
    s, c = np.zeros_like(x), np.zeros_like(x)
    O.lib().oracle_sincosf(x, s, c, x.size)
    assert s[0] == 0 and c[0] == 1 and s[1] == 0 and c[1] == 1
    assert np.isnan(s[2:5]).all() and np.isnan(c[2:5]).all()
    assert s[5] == x[5] and c[5] == 1
    assert abs(s[6]) <= 1 and abs(c[6]) <= 1


def philox_ref(c, k):
    """Independent pure-Python Philox4x32-10 (Salmon et al. 2011)."""
    c = list(c)
    k = list(k)
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + 0x9E3779B9) & 0xFFFFFFFF, (k[1] + 0xBB67AE85) & 0xFFFFFFFF]
    return c


def test_philox_known_answers(oracle):
    # Random123 known-answer vectors for philox4x32-10
    assert philox_ref([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert philox_ref([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert philox_ref([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    thr, steer, word = C.c_float(), C.c_float(), C.c_uint32()
    for seed, agent, step in [(1234, 0, 0), (1234, 4095, 1999), (7, 123456, 99), (0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF)]:
        O.lib().oracle_philox(seed, agent, step, C.byref(thr), C.byref(steer), C.byref(word))
        r = philox_ref([agent, step, 0, 0], [seed, 0x6F6B656E])
        t = np.float32(r[0] >> 8) * np.float32(2.0 ** -24) * np.float32(100.0)
        st = np.float32(r[1] >> 8) * np.float32(2.0 ** -24) * np.float32(10.0) - np.float32(5.0)
        assert np.float32(thr.value) == t and np.float32(steer.value) == st and word.value == r[2]
        assert 0.0 <= thr.value < 100.0 and -5.0 <= steer.value < 5.0


def test_ga_output_threshold_equals_the_written_out_sigmoid(oracle):
    """GeneticAgent's `sigmoid(z) > 0.5`: the product decides it with one comparison (z >= 0x33C00001, okenv_math.h), the oracle
    evaluates 1 / (1 + expf(-z)) as Network.hpp:162-165 writes it.  Same decision for every float within 2^20 ulps of the
    threshold, for every binade, for the special values and for a random sample; and the threshold is where the header says."""
    thr_bits = 0x33C00001
    near = (thr_bits + np.arange(-(1 << 20), (1 << 20), dtype=np.int64)).astype(np.uint32).view(np.float32)
    binades = np.array([s * 2.0 ** e * m for e in range(-149, 128) for m in (1.0, 1.5, 1.9999999) for s in (1, -1)], dtype=np.float64)
    rng = np.random.default_rng(3)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 8.94e-8, 8.9407e-8, 1.2e-7, 6e-8, 1e-7], dtype=np.float32)
    z1 = np.concatenate([near, binades.astype(np.float32), special, rng.normal(0, 1, 200000).astype(np.float32),
                         rng.normal(0, 1e-7, 200000).astype(np.float32)])
    z1 = z1[: (z1.size // 6) * 6]
    z = np.ascontiguousarray(z1.reshape(-1, 6))
    n = z.shape[0]
    to, so, tp, sp = (np.zeros(n, dtype=np.float32) for _ in range(4))
    O.lib().oracle_ga_decode(z, n, to, so, tp, sp)
    assert np.array_equal(to.view(np.uint32), tp.view(np.uint32)) and np.array_equal(so.view(np.uint32), sp.view(np.uint32))
    # the threshold itself, output 3 (steer +4) alone
    probe = np.zeros((3, 6), dtype=np.float32)
    probe[:, 3] = np.array([thr_bits - 1, thr_bits, thr_bits + 1], dtype=np.uint32).view(np.float32)
    probe[:, [0, 1, 2, 4, 5]] = -1.0
    to, so, tp, sp = (np.zeros(3, dtype=np.float32) for _ in range(4))
    O.lib().oracle_ga_decode(probe, 3, to, so, tp, sp)
    assert so.tolist() == [0.0, 4.0, 4.0] and sp.tolist() == [0.0, 4.0, 4.0]


def test_tanh_against_fp64_and_glibc(oracle):
    """ok_tanhf (include/okenv_math.h, the CMA-ES controller's activation): <= 1 ulp from the correctly rounded value (and
    equal to it on all but a handful of a million samples), within 2 ulps of glibc's tanhf (which is not correctly rounded
    itself); odd; saturates; NaN stays NaN."""
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-12, 12, 400000), rng.normal(0, 1, 400000), rng.uniform(-1e-3, 1e-3, 100000),
                        10.0 ** rng.uniform(-8, 1.5, 100000), np.linspace(-0.001, 0.001, 20001)]).astype(np.float32)
    t = np.zeros_like(x)
    O.lib().oracle_tanhf(x, t, x.size)
    ref = np.tanh(x.astype(np.float64)).astype(np.float32)
    d = ulp_diff(t, ref)
    assert d.max() <= 1 and (d != 0).mean() < 1e-4
    libm = C.CDLL("libm.so.6")
    libm.tanhf.restype, libm.tanhf.argtypes = C.c_float, [C.c_float]
    sub = x[::50]
    g = np.array([libm.tanhf(float(v)) for v in sub], dtype=np.float32)
    assert ulp_diff(t[::50], g).max() <= 2
    tm = np.zeros_like(x)
    O.lib().oracle_tanhf(-x, tm, x.size)
    assert np.array_equal(tm.view(np.uint32), (-t).view(np.uint32))
    sp = np.array([0.0, -0.0, 1e-30, 2.0 ** -12, 9.0, 10.0, 25.0, 1e30, np.inf, -np.inf, np.nan], dtype=np.float32)
    ts = np.zeros_like(sp)
    O.lib().oracle_tanhf(sp, ts, sp.size)
    assert ts[0] == 0 and np.signbit(ts[1]) and ts[2] == sp[2] and ts[5] == 1 and ts[6] == 1 and ts[7] == 1 and ts[8] == 1 and ts[9] == -1
    assert np.isnan(ts[10]) and abs(ts[4] - np.tanh(9.0)) < 1e-7
