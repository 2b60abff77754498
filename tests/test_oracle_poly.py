"""CPU: the ploidy-4 (autotetraploid) oracle is pinned to trajectories generated from the REAL reference
(poly_geno.c sweeps called by oracle/ref_dump_poly.c): byte-identical dumps, including the float genotype
frequency tables, the imputed genotypes and the seeds after every sweep."""
import os
import subprocess

import pytest

import golden_util as gu
import orc

DUMP = os.path.join(orc.ORC_DIR, "orc_dump_poly")
POLY = gu.make_golden.POLY_CASES


@pytest.fixture(scope="module", autouse=True)
def _build():
    orc.build()


@pytest.mark.parametrize("name", sorted(POLY))
def test_tetraploid_reference_configuration_is_byte_identical(name, tmp_path):
    N, L, K, A, miss, u, b, t, e, r, j, seeds = POLY[name]
    out = str(tmp_path / (name + ".out"))
    args = [DUMP, os.path.join(gu.GOLDEN, name + ".txt"), out] + [str(x) for x in (K, N, L, u, b, t, e, r, j) + tuple(seeds)]
    assert subprocess.call(args) == 0
    with open(out, "rb") as a, open(os.path.join(gu.GOLDEN, name + ".golden"), "rb") as g:
        assert a.read() == g.read()


ALLO = gu.make_golden.ALLO_CASES


@pytest.mark.parametrize("name", sorted(ALLO))
def test_allotetraploid_reference_configuration_is_byte_identical(name, tmp_path):
    """-ap 0 (update_P_allo, calc_exfreq_allo, allo_genfreq, choose_*_allo): tests/golden/ta*.golden come from the
    reference's own sweeps (oracle/ref_dump_poly.c ... 0)"""
    N, L, K, A, miss, u, b, t, e, r, j, seeds = ALLO[name]
    out = str(tmp_path / (name + ".out"))
    args = [DUMP, os.path.join(gu.GOLDEN, name + ".txt"), out] + [str(x) for x in (K, N, L, u, b, t, e, r, j) + tuple(seeds)] + ["0", "0", "0", "1"]
    assert subprocess.call(args) == 0
    with open(out, "rb") as a, open(os.path.join(gu.GOLDEN, name + ".golden"), "rb") as g:
        assert a.read() == g.read()


@pytest.mark.parametrize("name", sorted(ALLO))
def test_allotetraploid_canonical_configuration_keeps_the_discrete_trajectory(name, tmp_path):
    """math = ISG, exact sums: imputed genotypes, Z, counts, MH states and seeds identical after every sweep, doubles within 1e-9"""
    N, L, K, A, miss, u, b, t, e, r, j, seeds = ALLO[name]
    out = str(tmp_path / (name + ".can"))
    args = [DUMP, os.path.join(gu.GOLDEN, name + ".txt"), out] + [str(x) for x in (K, N, L, u, b, t, e, r, j) + tuple(seeds)] + ["1", "1", "0", "1"]
    assert subprocess.call(args) == 0
    got, want = gu.parse(out), gu.parse(os.path.join(gu.GOLDEN, name + ".golden"))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        fg, fw = gu.fields(g), gu.fields(w)
        for key in ("hz", "hgeno", "hcnt", "hqqnum", "seeds"):
            assert fg.get(key) == fw.get(key), (key, g, w)
        assert [t for t in g.split() if t.startswith("st")] == [t for t in w.split() if t.startswith("st")]
        a, bb = gu.floats(g), gu.floats(w)
        assert len(a) == len(bb)
        for x, y in zip(a, bb):
            assert x == y or (x != x and y != y) or abs(x - y) <= 1e-9 * max(abs(x), abs(y)), (g, w)


def test_tetraploid_coding_matches_numpy_restatement():
    import numpy as np
    from instruct_amd import synth
    for name in sorted(POLY):
        raw = gu.make_golden.poly_data_for(name)
        obs, alleleid, allelenum = synth.code_tetraploid(raw)
        line = [l for l in gu.parse(os.path.join(gu.GOLDEN, name + ".golden")) if l.startswith("data ")][0]
        f = gu.fields(line)
        assert orc.fnv_i32(obs) == f["hobs"] and orc.fnv_i32(alleleid) == f["halleleid"] and orc.fnv_i32(allelenum) == f["hallelenum"]
