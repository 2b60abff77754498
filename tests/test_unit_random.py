"""CPU: RNG and sampler known-answer vectors produced by the reference's random.c (oracle/ref_unit.c)."""
import os

import numpy as np
import pytest

import golden_util as gu
import orc


@pytest.fixture(scope="module")
def vectors():
    orc.build()
    recs = {}
    for line in gu.parse(os.path.join(gu.GOLDEN, "unit_random.golden")):
        recs.setdefault(line.split()[0], []).append(line.split()[1:])
    return recs


def _chain(math=0):
    geno = np.zeros((1, 1, 2), dtype=np.int32)
    return orc.OrcChain(geno, np.array([2], dtype=np.int32), np.zeros((1, 1), dtype=np.int32), 2, math=math)


def fh(x):
    return float.fromhex(x)


def test_ran1_sequence_and_seeds(vectors):
    c = _chain()
    c.setseeds(13, 4, 1972)
    for i, (idx, val) in enumerate(vectors["ran1"]):
        assert int(idx) == i and c.ran1() == fh(val)
    assert c.seeds() == tuple(int(x) for x in vectors["seeds"][0])


def test_samplers_in_stream_order(vectors):
    """rgamma / rgeom / rnormal / rdirich / disc_unif: same values AND same stream consumption."""
    c = _chain()
    c.setseeds(101, 202, 303)
    for a, g, s1, s2, s3 in vectors["rgamma"]:
        assert c.lib.orc_rgamma(c.h, fh(a)) == fh(g)
        assert c.seeds() == (int(s1), int(s2), int(s3))
    for p, g, s1, s2, s3 in vectors["rgeom"]:
        assert c.ran1() == fh(p)
        assert c.lib.orc_rgeom(c.h, fh(p)) == int(g)
        assert c.seeds() == (int(s1), int(s2), int(s3))
    for m, x, s1, s2, s3 in vectors["rnormal"]:
        mean = c.ran1() * 10
        assert mean == fh(m)
        assert c.lib.orc_rnormal(c.h, mean, 1.0) == fh(x)
        assert c.seeds() == (int(s1), int(s2), int(s3))
    for i, rec in enumerate(vectors["rdirich"]):
        n = int(rec[0])
        if i % 3 != 0:
            assert c.ran1() * 10 == fh(rec[1])
        add = fh(rec[1])
        alpha = np.array([fh(v) for v in rec[2:2 + n]])
        out = np.zeros(n)
        c.lib.orc_rdirich(c.h, orc._ptr(alpha), n, orc._ptr(out), add)
        assert [fh(v) for v in rec[3 + n:3 + 2 * n]] == list(out)
        assert c.seeds() == tuple(int(v) for v in rec[3 + 2 * n:])
    for i, rec in enumerate(vectors["disc_unif"]):
        n = int(rec[0])
        vec = np.zeros(n)
        acc = 0.0
        for k in range(n):
            w = 0.0 if (i + k) % 5 == 0 else c.ran1()
            acc += w
            vec[k] = acc
        if acc == 0:
            vec[n - 1] = 1.0
        assert [fh(v) for v in rec[1:1 + n]] == list(vec)
        assert c.lib.orc_disc_unif(c.h, orc._ptr(vec), n) == int(rec[2 + n])
        assert c.seeds() == tuple(int(v) for v in rec[3 + n:])


def test_canonical_samplers_keep_stream_positions(vectors):
    """With isg_math the accept/reject decisions (stream consumption) stay those of the reference."""
    c = _chain(math=1)
    c.setseeds(101, 202, 303)
    for a, g, s1, s2, s3 in vectors["rgamma"]:
        v = c.lib.orc_rgamma(c.h, fh(a))
        assert abs(v - fh(g)) <= 1e-12 * abs(fh(g))
        assert c.seeds() == (int(s1), int(s2), int(s3))
