"""GPU (pytest -m gpu): replay update_P on the device (walk engine, instruct_amd/csrc/isg_walk_hip.inc) -- the allele frequencies
and the stream position after every update_P equal the canonical oracle's (which draws the Dirichlets one after the other as
mcmc.c:846-857 / random.c:233-280 do), with the sweeps really taken by the device path."""
import numpy as np
import pytest

import orc
from instruct_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _libs():
    orc.build()
    capi.load()


def _pair(geno, an, mi, K, seeds=(13, 4, 1972)):
    h = capi.HipChain(geno, an, mi, K, rng_sched=capi.SCHED_REPLAY)
    o = orc.OrcChain(geno, an, mi, K, math=orc.MATH_ISG, accum=orc.ACC_EXACT, sched=orc.SCHED_REPLAY)
    h.setseeds(*seeds)
    o.setseeds(*seeds)
    initd = np.array([h.ran1() for _ in range(K)], dtype=np.float32)
    [o.ran1() for _ in range(K)]
    h.chain_init(initd)
    o.chain_init(initd)
    return h, o


CASES = [
    # N, L, K, missing, alleles
    (50, 100, 3, 0.0, 2),
    (40, 700, 4, 0.05, 4),     # rare alleles: shapes 1 (one uniform), 2, 3 .. mixed with large ones
    (300, 3000, 5, 0.02, 2),   # several blocks and super-blocks
    (24, 5000, 9, 0.0, 3),
]


@pytest.mark.parametrize("case", CASES)
def test_update_P_on_the_device_equals_the_sequential_draws(case):
    N, L, K, miss, nall = case
    geno, an, mi = synth.code_diploid(synth.raw_alleles(N, L, K, 2, nall, miss, 11))
    h, o = _pair(geno, an, mi, K)
    for it in range(4):
        h.update_P()
        o.update_P()
        assert np.array_equal(h.freq(), o.freq()), (case, it)
        assert h.seeds() == o.seeds(), (case, it)
        for f in ("update_S_POP", "update_G", "update_ZQ", "update_alpha", "cal_lkh"):
            getattr(h, f)()
            getattr(o, f)()
        assert h.seeds() == o.seeds(), (case, it)
    st = h.p_device_stats()
    assert st["device_sweeps"] == 4 and st["host_sweeps"] == 0, st
    h.close()


def test_update_P_device_at_config_2_size_and_with_small_segments(monkeypatch):
    geno, an, mi = synth.make_diploid(2000, 1000, 5)
    for seg in (None, "1024"):
        if seg:
            monkeypatch.setenv("INSTRUCT_WALK_SEG", seg)
        h, o = _pair(geno, an, mi, 5)
        for it in range(3):
            h.iteration()
            o.iteration()
            assert np.array_equal(h.freq(), o.freq()) and h.seeds() == o.seeds(), (seg, it)
        st = h.p_device_stats()
        assert st["device_sweeps"] == 3 and st["host_sweeps"] == 0, st
        if seg:
            assert st["segments"] == 5
        h.close()


def test_a_missed_window_falls_back_to_the_host_loop_with_the_same_values(monkeypatch):
    monkeypatch.setenv("INSTRUCT_WALK_K", "1.0")  # windows of one sigma: misses are certain at this size
    geno, an, mi = synth.make_diploid(400, 2500, 5)
    h, o = _pair(geno, an, mi, 5)
    for it in range(3):
        h.iteration()
        o.iteration()
        assert np.array_equal(h.freq(), o.freq()) and h.seeds() == o.seeds(), it
    st = h.p_device_stats()
    assert st["host_sweeps"] >= 1, st
    h.close()


def _iterate_and_compare(h, o, n, tag):
    for it in range(n):
        h.iteration()
        o.iteration()
        assert np.array_equal(h.z(), o.z()), (tag, it)
        assert np.array_equal(h.qq(), o.qq()) and np.array_equal(h.qqnum(), o.qqnum()), (tag, it)
        assert h.seeds() == o.seeds() and h.alpha() == o.alpha() and h.totallkh() == o.totallkh(), (tag, it)


@pytest.mark.parametrize("shape", [(600, 4000, 5), (250, 6000, 10), (300, 5000, 3)])
def test_update_ZQ_by_shape_intervals_settles_and_equals_the_oracle(shape):
    """large clusters (early in a chain): the interval resolver settles the sweeps itself; K = 10 has no block resolver behind it"""
    N, L, K = shape
    geno, an, mi = synth.make_diploid(N, L, K)
    h, o = _pair(geno, an, mi, K)
    _iterate_and_compare(h, o, 4, shape)
    st = h.zq_spec_stats()
    assert st["settled"] >= 2 and h.zq_fallbacks() == 0, st
    h.close()


def test_update_ZQ_by_shape_intervals_hands_small_clusters_on(monkeypatch):
    """N = 50, L = 100: every shape is small, every candidate uncertain -- the sweep must come out right whoever settles it"""
    geno, an, mi = synth.code_diploid(synth.raw_alleles(50, 100, 3, 2, 2, 0.05, 7))
    h, o = _pair(geno, an, mi, 3)
    _iterate_and_compare(h, o, 6, "small")
    st = h.zq_spec_stats()
    assert st["tried"] >= 1, st
    h.close()


def test_update_ZQ_by_shape_intervals_lost_sweep_is_redone_by_the_next_path(monkeypatch):
    monkeypatch.setenv("INSTRUCT_ZQ_SPEC_TEST_ABORT", "2")
    geno, an, mi = synth.make_diploid(400, 3000, 5)
    h, o = _pair(geno, an, mi, 5)
    _iterate_and_compare(h, o, 4, "abort")
    st = h.zq_spec_stats()
    assert st["lost"] >= 1, st
    h.close()


def test_update_ZQ_by_shape_intervals_with_several_segments_and_a_narrow_band(monkeypatch):
    monkeypatch.setenv("INSTRUCT_ZQ_SPEC_SEG", "256")
    monkeypatch.setenv("INSTRUCT_ZQ_SPEC_BAND", "6")
    geno, an, mi = synth.make_diploid(1500, 3000, 5)
    h, o = _pair(geno, an, mi, 5)
    _iterate_and_compare(h, o, 3, "segments")
    st = h.zq_spec_stats()
    assert st["segments"] >= 5 and st["tried"] >= 1, st
    h.close()


@pytest.mark.parametrize("K", [9, 10, 12, 16])
def test_update_ZQ_start_positions_resolved_for_K_beyond_8(K):
    """K = 9 .. 16 (the block resolver stops at 8): the interval resolver settles the sweeps, nothing goes to the chain kernels"""
    geno, an, mi = synth.make_diploid(160, 1000 * K, K)
    h, o = _pair(geno, an, mi, K)
    _iterate_and_compare(h, o, 3, K)
    st = h.zq_spec_stats()
    assert st["settled"] == 3 and st["lost"] == 0 and h.zq_fallbacks() == 0, st
    h.close()
