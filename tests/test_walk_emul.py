"""The walk engine (instruct_amd/csrc/isg_walk.h) under host emulation: the kernel bodies the device compiles, run workgroup by
workgroup on the CPU, must give every Dirichlet of a run the start position the sequential sampler (random.c:233-280) reaches."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL = os.path.join(ROOT, "tests", "emul")
LIB = os.path.join(EMUL, "libwalk_emul.so")
SRCS = [os.path.join(EMUL, "walk_emul.cpp")] + [os.path.join(ROOT, "instruct_amd", "csrc", f) for f in
                                                ("isg_walk.h", "isg_walk_plan.h", "isg_sampler.h", "isg_wh.h", "isg_math.h")]


def lib():
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SRCS):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-o", LIB, SRCS[0]])
    l = C.CDLL(LIB)
    l.wk_emul_expected_rej.restype = C.c_float
    l.wk_emul_expected_rej.argtypes = [C.c_float]
    return l


def run_exact(gam0, cnt, seeds=(13, 4, 1972), rho_hi=0.6, sigma=1.0, kwin=5.0, seg_groups=24576, scale=1.0, nthreads=256):
    l = lib()
    G = len(gam0) - 1
    gam0 = np.ascontiguousarray(gam0, dtype=np.int32)
    cnt = np.ascontiguousarray(cnt, dtype=np.int32)
    T = np.zeros(G + 1, dtype=np.uint64)
    out = np.zeros(8, dtype=np.uint64)
    rc = l.wk_emul_exact(gam0.ctypes, G, cnt.ctypes, C.c_long(seeds[0]), C.c_long(seeds[1]), C.c_long(seeds[2]), C.c_double(rho_hi), C.c_double(sigma), C.c_double(kwin),
                         seg_groups, C.c_float(scale), nthreads, T.ctypes, out.ctypes)
    Tr = np.zeros(G + 1, dtype=np.uint64)
    tot = C.c_ulonglong(0)
    l.wk_ref_exact(gam0.ctypes, G, cnt.ctypes, C.c_long(seeds[0]), C.c_long(seeds[1]), C.c_long(seeds[2]), Tr.ctypes, C.byref(tot))
    return rc, T, out, Tr, tot.value


def groups_of(sizes):
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)


@pytest.mark.parametrize("G,A,seed", [(300, 2, 1), (2000, 2, 2), (1500, 4, 3), (5000, 3, 4)])
def test_exact_run_equals_sequential_sampler(G, A, seed):
    rng = np.random.default_rng(seed)
    gam0 = groups_of(np.full(G, A))
    cnt = rng.integers(0, 4000, size=G * A)
    rc, T, out, Tr, tot = run_exact(gam0, cnt, seeds=(13 + seed, 4, 1972))
    assert rc == 0 and out[1] == 0, out
    assert np.array_equal(T, Tr)
    assert out[0] == tot


def test_small_and_zero_counts_shift_the_windows():
    """shapes 1 (one uniform, no rejection), 2 (no u1 transform), 3..10 (most rejections): the device-side window centres follow the shapes"""
    rng = np.random.default_rng(7)
    sizes = rng.integers(2, 7, size=3000)
    gam0 = groups_of(sizes)
    cnt = rng.choice([0, 0, 1, 1, 2, 3, 5, 9, 40, 3000], size=int(sizes.sum()))
    rc, T, out, Tr, tot = run_exact(gam0, cnt, rho_hi=0.8, sigma=1.2)
    assert rc == 0 and out[1] == 0, out
    assert np.array_equal(T, Tr) and out[0] == tot


def test_several_segments_and_small_blocks():
    rng = np.random.default_rng(11)
    G = 6000
    gam0 = groups_of(np.full(G, 2))
    cnt = rng.integers(50, 9000, size=2 * G)
    for seg in (1024, 2048, 24576):
        rc, T, out, Tr, tot = run_exact(gam0, cnt, seg_groups=seg)
        assert rc == 0 and out[1] == 0, (seg, out)
        assert np.array_equal(T, Tr) and out[0] == tot
        assert out[4] == -(-G // seg)


def test_window_miss_is_reported_not_silently_wrong():
    rng = np.random.default_rng(5)
    G = 4000
    gam0 = groups_of(np.full(G, 2))
    cnt = rng.integers(50, 9000, size=2 * G)
    rc, T, out, Tr, tot = run_exact(gam0, cnt, sigma=0.05, kwin=1.0)  # windows far too narrow
    assert rc == 0 and out[1] != 0


def test_expected_rejections_table():
    l = lib()
    assert l.wk_emul_expected_rej(1.0) == 0.0
    assert abs(l.wk_emul_expected_rej(2.0) - 0.4102) < 1e-3
    assert abs(l.wk_emul_expected_rej(3.0) - 0.7117) < 1e-3
    assert abs(l.wk_emul_expected_rej(1e6) - 0.4841) < 1e-3


def test_interval_tables_certain_bytes_are_exact_for_every_shape_in_the_interval():
    """interval mode (update_ZQ): a byte without the UNCERTAIN flag must be the consumption of the sequential sampler for ANY shapes
    inside the intervals; flagged bytes are allowed to be wrong (they get probed)"""
    l = lib()
    l.wk_ref_consumed.restype = C.c_uint
    rng = np.random.default_rng(21)
    N, K, nv = 900, 5, 700
    seeds = (13, 4, 1972)
    gam0 = (np.arange(N + 1) * K).astype(np.int32)
    B = np.arange(N + 1, dtype=np.uint64) * np.uint64(2 * nv + 2 * K)
    gpos = np.zeros(N * K + 1, dtype=np.uint64)
    for m in range(K):
        gpos[m:N * K:K] = B[:N] + np.uint64(2 * nv + 2 * m)
    gpos[N * K] = B[N]
    alpha = 3.7
    lam = rng.integers(150, 5000, size=(N, K)).astype(np.float64)
    half = np.ceil(4.5 * np.sqrt(lam * 0.8)) + 1
    alo = (lam - half + alpha).astype(np.float32).reshape(-1)
    ahi = (lam + half + alpha).astype(np.float32).reshape(-1)
    T = np.zeros(N + 1, dtype=np.uint64)
    path = np.zeros(N, dtype=np.uint8)
    out = np.zeros(8, dtype=np.uint64)
    rc = l.wk_emul_interval(gam0.ctypes, N, gpos.ctypes, alo.ctypes, ahi.ctypes, C.c_long(seeds[0]), C.c_long(seeds[1]), C.c_long(seeds[2]), C.c_double(0.8), C.c_double(1.0),
                            C.c_double(5.0), 4096, C.c_float(1.0), 256, 0, T.ctypes, path.ctypes, out.ctypes)
    assert rc == 0 and out[0] == 0, out
    nunc = int((path & 0x80 != 0).sum())
    assert 0 < nunc < N // 4, nunc            # a few per cent of the candidates at these shapes
    checked = 0
    for i in range(N):
        if path[i] & 0x80 or path[i] == 0xff:
            continue
        for trial in range(3):                # the interval's ends and a random interior point
            f = (0.0, 1.0, rng.random())[trial]
            shapes = np.ascontiguousarray((lam[i] - half[i] + alpha) + f * 2 * half[i], dtype=np.float64)
            used = l.wk_ref_consumed(C.c_ulonglong(int(gpos[i * K]) + 2 * int(T[i])), shapes.ctypes, K, C.c_long(seeds[0]), C.c_long(seeds[1]), C.c_long(seeds[2]))
            assert used == 2 * K + 2 * int(path[i] & 0x7f), (i, trial, used, path[i])
            checked += 1
    assert checked > 1500


def _spec_problem(N, K, nv, seed, lo=150, hi=5000):
    rng = np.random.default_rng(seed)
    gam0 = (np.arange(N + 1) * K).astype(np.int32)
    B = np.arange(N + 1, dtype=np.uint64) * np.uint64(2 * nv + 2 * K)
    gpos = np.zeros(N * K + 1, dtype=np.uint64)
    for m in range(K):
        gpos[m:N * K:K] = B[:N] + np.uint64(2 * nv + 2 * m)
    gpos[N * K] = B[N]
    alpha = 3.7
    lam = rng.integers(lo, hi, size=(N, K)).astype(np.float64)
    half = np.ceil(4.5 * np.sqrt(lam * 0.8)) + 1
    alo = (lam - half + alpha).astype(np.float32).reshape(-1)
    ahi = (lam + half + alpha).astype(np.float32).reshape(-1)
    atrue = np.ascontiguousarray((lam + np.round((rng.random((N, K)) * 2 - 1) * (half - 1)) + alpha).reshape(-1), dtype=np.float64)
    return gam0, gpos, alo, ahi, atrue


@pytest.mark.parametrize("seg,band", [(4096, 24), (256, 24), (320, 40)])
def test_interval_resolver_end_to_end_lenient_walk_probes_strict_walk(seg, band):
    """several segments: the strict walk enters the later ones a little off the lenient walk's entry (its tables' origin)"""
    l = lib()
    N, K = 1200, 5
    gam0, gpos, alo, ahi, atrue = _spec_problem(N, K, 600, 31)
    T = np.zeros(N + 1, dtype=np.uint64)
    Tref = np.zeros(N + 1, dtype=np.uint64)
    out = np.zeros(8, dtype=np.uint64)
    rc = l.wk_emul_spec(gam0.ctypes, N, gpos.ctypes, alo.ctypes, ahi.ctypes, atrue.ctypes, C.c_long(13), C.c_long(4), C.c_long(1972), C.c_double(1.0), C.c_double(5.0), seg, band,
                        2 * band + 32, 256, T.ctypes, Tref.ctypes, out.ctypes)
    assert rc == 0 and out[0] == 0, out
    assert out[1] == 0, out          # the strict walk got through ...
    assert np.array_equal(T, Tref)   # ... and its trajectory is the sequential sampler's
    assert out[2] > 0 and out[3] == -(-N // seg)


def test_interval_resolver_reports_when_the_band_is_too_narrow():
    l = lib()
    N, K = 1500, 5
    gam0, gpos, alo, ahi, atrue = _spec_problem(N, K, 400, 5, lo=60, hi=400)   # smaller shapes: many uncertain bytes, many wrong guesses
    T = np.zeros(N + 1, dtype=np.uint64)
    Tref = np.zeros(N + 1, dtype=np.uint64)
    out = np.zeros(8, dtype=np.uint64)
    rc = l.wk_emul_spec(gam0.ctypes, N, gpos.ctypes, alo.ctypes, ahi.ctypes, atrue.ctypes, C.c_long(13), C.c_long(4), C.c_long(1972), C.c_double(1.0), C.c_double(5.0), 4096, 1,
                        34, 256, T.ctypes, Tref.ctypes, out.ctypes)
    assert rc == 0
    assert out[1] != 0 or np.array_equal(T, Tref)   # either it says so, or it is right: never silently wrong
