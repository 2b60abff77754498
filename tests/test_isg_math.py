"""CPU: the bit-reproducible math header shared by the canonical oracle mode and the HIP kernels."""
import ctypes as C
import math

import numpy as np
import pytest

import orc


@pytest.fixture(scope="module")
def lib():
    orc.build()
    lib = orc.load()
    for f in ("orc_isg_log", "orc_isg_exp", "orc_isg_cos"):
        getattr(lib, f).restype = C.c_double
        getattr(lib, f).argtypes = [C.c_double]
    lib.orc_isg_pow.restype = C.c_double
    lib.orc_isg_pow.argtypes = [C.c_double, C.c_double]
    lib.orc_isg_accsum.restype = C.c_double
    lib.orc_isg_accsum.argtypes = [C.c_void_p, C.c_long]
    return lib


def ulps(a, b):
    if a == b:
        return 0
    ia, ib = np.float64(a).view(np.int64), np.float64(b).view(np.int64)
    return abs(int(ia) - int(ib))


def test_log_exp_within_one_ulp_of_libm(lib):
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.random(20000), rng.random(20000) * 1e-6, np.exp((rng.random(20000) - 0.5) * 1400),
                         1 + (rng.random(20000) - 0.5) * 1e-3])
    assert max(ulps(lib.orc_isg_log(float(x)), math.log(x)) for x in xs) <= 1
    ys = (rng.random(40000) - 0.5) * 1400
    assert max(ulps(lib.orc_isg_exp(float(y)), math.exp(y)) for y in ys) <= 1


def test_pow_accuracy_on_sampler_ranges(lib):
    rng = np.random.default_rng(2)
    worst = 0
    for _ in range(20000):  # rgamma1 / proposal ranges: moderate exponents
        x, y = rng.random() * 20, rng.random() * 60 - 10
        worst = max(worst, ulps(lib.orc_isg_pow(x, y), math.pow(x, y)))
    assert worst <= 2
    worst = 0
    for _ in range(20000):  # update_alpha: q in (0,1), exponent up to 2*L
        x, y = rng.random(), rng.random() * 20000
        r = math.pow(x, y)
        if r > 1e-300:
            worst = max(worst, ulps(lib.orc_isg_pow(x, y), r))
    assert worst <= 64


def test_special_values(lib):
    assert lib.orc_isg_log(0.0) == -math.inf and math.isnan(lib.orc_isg_log(-1.0))
    assert lib.orc_isg_exp(-800.0) == 0.0 and lib.orc_isg_exp(800.0) == math.inf
    assert lib.orc_isg_pow(0.0, 0.0) == 1.0 and lib.orc_isg_pow(0.0, 3.0) == 0.0 and lib.orc_isg_pow(0.3, 0.0) == 1.0
    assert lib.orc_isg_pow(0.5, 49.0) == 2.0 ** -49
    assert lib.orc_isg_exp(-740.0) == math.exp(-740.0)  # subnormal result
    ts = np.linspace(0, 2 * 3.141592654, 5000)
    assert max(abs(lib.orc_isg_cos(float(t)) - math.cos(t)) for t in ts) < 3e-16


def test_accumulator_is_order_independent_and_accurate(lib):
    rng = np.random.default_rng(3)
    v = np.log(rng.random(50000)) * (1 + rng.integers(0, 7, 50000))
    a = lib.orc_isg_accsum(orc._ptr(v), len(v))
    w = v[rng.permutation(len(v))].copy()
    assert lib.orc_isg_accsum(orc._ptr(w), len(w)) == a
    assert abs(a - math.fsum(v)) <= 2e-16 * abs(a)
    inf = np.array([1.0, -np.inf, 2.0])
    assert lib.orc_isg_accsum(orc._ptr(inf), 3) == -math.inf
