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
