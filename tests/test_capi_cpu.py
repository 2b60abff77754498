"""CPU: the C-ABI library loads, exports every symbol declared in include/instruct_hip.h, and the
product path fails loudly when no MI355X is present (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from instruct_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIB_PATH):
        subprocess.check_call(["python", "-m", "instruct_amd.build"], cwd=ROOT)
    return capi.load()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "instruct_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(isg_\w+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), n
    assert set(capi.EXPORTS) <= set(names)


def test_gelman_rubin_entry_point_matches_reference_value(lib):
    import golden_util as gu
    lines = gu.parse(os.path.join(gu.GOLDEN, "c1_c2.golden"))
    convg = gu.floats([l for l in lines if l.startswith("convg")][0])
    gr = gu.floats([l for l in lines if l.startswith("GR")][0])[0]
    c = gu.case_args("c1_c2")
    assert capi.gelman_rubin(convg, c["c"], c["r"]) == gr


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="a GPU is present")
def test_no_gpu_means_loud_failure(lib):
    geno = np.zeros((4, 8, 2), dtype=np.int32)
    with pytest.raises(capi.IsgError, match="no HIP device"):
        capi.HipChain(geno, np.full(8, 2, dtype=np.int32), np.zeros((4, 8), dtype=np.int32), 2)


def test_shim_object_compiles_against_the_abi():
    from instruct_amd import build
    obj = build.build_host(force=True)
    out = subprocess.check_output(["nm", "-g", "--defined-only", obj]).decode()
    for sym in ("mcmc_updating", "free_chain", "chcksame", "genofreq_inbreedcoff", "dgeom", "print_info", "adpt_indp",
                "hastings_stat", "dt_stat", "allocate_node", "free_node", "allocate_chn", "store_chn", "check_empty_cluster"):
        assert re.search(r"\bT %s\b" % sym, out), sym  # the exported surface of reference mcmc.h:56-69


def test_fast_tetraploid_coder_equals_the_restatement_of_transform_data2():
    """synth.code_tetraploid_fast builds every large ploidy-4 input (bench.py, the config-5 tests, the profiling tools);
    synth.code_tetraploid restates transform_data2 (data_interface.c:571-669) and is what the golden cases are coded with"""
    import numpy as np
    from instruct_amd import synth
    for (N, L, K, A, miss, seed) in ((40, 60, 3, 4, 0.05, 1), (25, 30, 2, 6, 0.2, 2), (12, 50, 4, 2, 0.0, 3), (9, 20, 3, 8, 0.5, 4)):
        raw = synth.raw_alleles(N, L, K, 4, A, miss, seed)
        if N == 9:
            raw[:, 3, :] = synth.MISSING   # a locus nobody was typed at
        a, b = synth.code_tetraploid(raw), synth.code_tetraploid_fast(raw)
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and np.array_equal(x, y)


@pytest.mark.parametrize("units,mu,sigma,a,shape", [(512, 2.42, 1.89, 1.5, 1.2), (256, 2.42, 1.89, 1.8, 0.0), (16, 2.3, 1.8, 1.5, 1.2),
                                                    (512, 3.7, 2.3, 0.6, 0.0), (512, 2.4, 1.9, 12.0, 0.0), (504, 0.9, 1.0, 1.5, 1.2), (512, 40.0, 9.0, 1.5, 1.2)])
def test_resolver_plan_invariants(lib, units, mu, sigma, a, shape):
    """The block resolver's plan (host logic of instruct_amd/csrc/isg_resolve_hip.inc): every (individual, candidate) of every
    window belongs to exactly one unit, table slots do not collide, the windows hold what the walk relies on."""
    import ctypes as C
    lo, w = np.zeros(64, np.int16), np.zeros(64, np.int16)
    ur, uo, cs = np.zeros(512, np.int16), np.zeros(512, np.int16), np.zeros(512, np.int16)
    nun = C.c_int(0)
    lib.isg_zq_resolve_plan.restype = C.c_int
    lib.isg_zq_resolve_plan.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double] + [C.c_void_p] * 5 + [C.POINTER(C.c_int)]
    D = lib.isg_zq_resolve_plan(units, mu, sigma, a, shape, lo.ctypes.data, w.ctypes.data, ur.ctypes.data, uo.ctypes.data, cs.ctypes.data, C.byref(nun))
    n = nun.value
    assert 1 <= D <= 64 and 1 <= n <= units
    assert lo[0] == 0 and w[0] == 4                       # the block's first individual starts at its known e: candidate 0 is evaluated
    assert all(w[r] % 4 == 0 and 4 <= w[r] <= 128 for r in range(D)) and all(lo[r] >= 0 for r in range(D))
    assert all(0 <= lo[r] - lo[r - 1] <= 100 for r in range(1, D))   # an irregular entry (255) must leave the next window
    assert n == sum(int(w[r]) // 4 for r in range(D))
    seen, slots = set(), set()
    for u in range(n):
        r, o = int(ur[u]), int(uo[u])
        assert 0 <= r < D and lo[r] <= o < lo[r] + w[r] and (o - lo[r]) % 4 == 0
        assert (r, o) not in seen
        seen.add((r, o))
        rel = o - int(lo[r])
        assert int(cs[u]) == (((r >> 2) * 128 + rel) << 2) + (r & 3)   # byte r % 4 of dword (r / 4) * 128 + rel; the other three at + 4, 8, 12
        for k in range(4):
            assert int(cs[u]) + 4 * k not in slots
            slots.add(int(cs[u]) + 4 * k)
    assert len(seen) == n
    # an individual's units are consecutive (they share its genotype row and tape stretch in one XCD's L2)
    for r in range(D):
        idx = [u for u in range(n) if ur[u] == r]
        assert idx == list(range(idx[0], idx[0] + len(idx)))
