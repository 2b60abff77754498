"""CPU: the oracle (CPU restatement) is pinned to trajectories generated from the REAL reference.

tests/golden/<case>.golden were written by oracle/ref_dump.c, which calls the reference's own
update_P / update_S_POP / update_G / update_ZQ / update_alpha / cal_lkh (see tests/golden/make_golden.py).
"""
import os
import subprocess

import pytest

import golden_util as gu
import orc

DUMP = os.path.join(orc.ORC_DIR, "orc_dump")


@pytest.fixture(scope="module", autouse=True)
def _build():
    orc.build()


@pytest.mark.parametrize("name", sorted(gu.CASES))
def test_reference_configuration_is_byte_identical(name, tmp_path):
    """libm + sequential sums + replay schedule == the reference, bit for bit, every sweep."""
    txt = gu.case_text(name, tmp_path)
    out = str(tmp_path / (name + ".out"))
    subprocess.check_call([DUMP] + gu.dump_cmd_args(name, txt, out))
    with open(out, "rb") as a, open(os.path.join(gu.GOLDEN, name + ".golden"), "rb") as b:
        assert a.read() == b.read()


@pytest.mark.parametrize("name", sorted(gu.CASES))
def test_canonical_configuration_keeps_the_discrete_trajectory(name, tmp_path):
    """isg_math + order-independent sums (what the GPU computes): Z hashes, allele counts, generations,
    qqnum and the RNG position are IDENTICAL to the reference at every sweep; doubles agree to 1e-9."""
    txt = gu.case_text(name, tmp_path)
    out = str(tmp_path / (name + ".out"))
    subprocess.check_call([DUMP] + gu.dump_cmd_args(name, txt, out) + ["1", "1", "0"])
    a = gu.parse(out)
    b = gu.parse(os.path.join(gu.GOLDEN, name + ".golden"))
    assert len(a) == len(b)
    for x, y in zip(a, b):
        fx, fy = gu.fields(x), gu.fields(y)
        for key in ("hz", "hcnt", "hgen", "hqqnum", "seeds", "hgeno", "hvalid"):
            if key in fy:
                assert fx.get(key) == fy[key], (key, x, y)
        vx, vy = gu.floats(x), gu.floats(y)
        assert len(vx) == len(vy)
        for p, q in zip(vx, vy):
            assert p == q or abs(p - q) <= 1e-9 * abs(q), (x, y)


def test_gelman_rubin_matches_reference_value():
    lines = gu.parse(os.path.join(gu.GOLDEN, "c1_c2.golden"))
    convg = gu.floats([l for l in lines if l.startswith("convg")][0])
    gr = gu.floats([l for l in lines if l.startswith("GR")][0])[0]
    c = gu.case_args("c1_c2")
    assert orc.gelman_rubin(convg, c["c"], c["r"]) == gr


def test_text_reader_matches_numpy_coding(tmp_path):
    import ctypes as C
    import numpy as np
    from instruct_amd import synth
    lib = orc.load()
    for name in ("c1_miss", "c1_a3"):
        path = os.path.join(gu.GOLDEN, name + ".txt")
        N, L = C.c_int(), C.c_int()
        an, g, m = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
        assert lib.orc_read_text_diploid(path.encode(), C.byref(N), C.byref(L), C.byref(an), C.byref(g), C.byref(m)) == 0
        geno, allelenum, miss = synth.code_diploid(synth.read_text_diploid(path))
        assert (N.value, L.value) == geno.shape[:2]
        assert np.array_equal(np.ctypeslib.as_array(an, (L.value,)), allelenum)
        assert np.array_equal(np.ctypeslib.as_array(g, (N.value, L.value, 2)), geno)
        assert np.array_equal(np.ctypeslib.as_array(m, (N.value, L.value)), miss)
