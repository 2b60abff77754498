"""GPU (pytest -m gpu): the HIP path, called through the C ABI, against
  (1) the canonical oracle configuration -- bit-exact on ALL state after every sweep, both RNG schedules;
  (2) the golden trajectories generated from the REAL reference -- bit-exact on Z, allele counts,
      generations, qqnum and the RNG seeds after every sweep; doubles within 1e-9; final posterior
      means (CHAIN) within the north-star tolerance 1e-6 relative;
  (3) size-independent properties at the full benchmark size (N=10000, L=5000, K=5).
"""
import os
import subprocess

import numpy as np
import pytest

import golden_util as gu
import orc
from instruct_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _libs():
    orc.build()
    capi.load()


def _pair(geno, an, mi, K, sched, mode=2, y=1, e=1, seeds=(13, 4, 1972)):
    h = capi.HipChain(geno, an, mi, K, mode=mode, type_freq=y, back_refl=e, rng_sched=sched)
    o = orc.OrcChain(geno, an, mi, K, mode=mode, type_freq=y, back_refl=e, math=orc.MATH_ISG, accum=orc.ACC_EXACT, sched=sched)
    h.setseeds(*seeds)
    o.setseeds(*seeds)
    initd = np.array([h.ran1() for _ in range(K)], dtype=np.float32)
    assert np.array_equal(initd, np.array([o.ran1() for _ in range(K)], dtype=np.float32))
    return h, o, initd


def _same(h, o, names, where):
    for n in names:
        a, b = getattr(h, n)(), getattr(o, n)()
        if isinstance(a, np.ndarray):
            assert np.array_equal(a, np.asarray(b)), (where, n)
        else:
            assert a == b, (where, n, a, b)


SWEEP_CASES = [
    # N, L, K, missing, alleles, mode, y, e
    (50, 100, 3, 0.0, 2, 2, 1, 1),
    (50, 100, 3, 0.05, 3, 2, 1, 1),
    (37, 1031, 5, 0.10, 2, 2, 1, 1),     # ragged: L not a multiple of the lane tile, heavy missing data
    (300, 1500, 5, 0.02, 2, 2, 1, 1),
    (64, 200, 4, 0.0, 2, 1, 1, 1),       # mode 1 (admixture only)
    (64, 200, 4, 0.03, 4, 2, 0, 0),      # -y 0, -e 0, four alleles
    (16, 9000, 9, 0.01, 2, 2, 1, 1),     # K > 8 kernel variant, several passes per row
    (5, 8, 2, 0.3, 2, 2, 1, 1),          # tiny, individuals with (almost) everything missing
    (50, 100, 3, 0.0, 2, 4, 1, 1),       # mode 4: population inbreeding coefficients
    (64, 700, 4, 0.05, 3, 4, 1, 0),      # mode 4, -e 0, three alleles, several workgroups
    (20, 60, 9, 0.02, 2, 4, 1, 1),       # mode 4, K > 8
    (50, 100, 3, 0.0, 2, 3, 1, 1),       # mode 3: one selfing rate per individual
    (40, 600, 4, 0.05, 3, 3, 0, 1),      # mode 3, -y 0, three alleles
    (50, 100, 3, 0.0, 2, 5, 1, 1),       # mode 5: one inbreeding coefficient per individual
    (30, 900, 5, 0.08, 4, 5, 1, 1),      # mode 5, four alleles, missing data
]


@pytest.mark.parametrize("sched", [capi.SCHED_REPLAY, capi.SCHED_KEYED])
@pytest.mark.parametrize("case", SWEEP_CASES)
def test_every_sweep_bit_exact_vs_canonical_oracle(case, sched):
    N, L, K, miss, nall, mode, y, e = case
    geno, an, mi = synth.code_diploid(synth.raw_alleles(N, L, K, 2, nall, miss, 7))
    h, o, initd = _pair(geno, an, mi, K, sched, mode, y, e)
    pos = ["seeds"] if sched == capi.SCHED_REPLAY else []
    h.chain_init(initd)
    o.chain_init(initd)
    _same(h, o, ["z", "qq", "qqnum", "generation", "alpha"] + (["self_rates"] if mode in (3, 5) else []) + pos, "init")
    import ctypes as C
    o.lib.orc_iter_advance.argtypes = [C.c_void_p]
    h.lib.isg_iter_advance.argtypes = [C.c_void_p]
    for it in range(3):
        h.update_P(); o.update_P()
        _same(h, o, ["count_alleles", "freq"] + pos, (it, "P"))
        if mode == 2:
            h.update_S_POP(); o.update_S_POP()
            _same(h, o, ["self_rates", "state"] + pos, (it, "S"))
            h.update_G(); o.update_G()
            _same(h, o, ["generation"] + pos, (it, "G"))
        if mode == 3:
            h.update_S_IND(); o.update_S_IND()
            _same(h, o, ["self_rates"] + pos, (it, "SI"))
            h.update_G(); o.update_G()
            _same(h, o, ["generation"] + pos, (it, "G"))
        if mode == 5:
            h.update_S_IND(); o.update_S_IND()   # update_F_IND
            _same(h, o, ["self_rates"] + pos, (it, "FI"))
        if mode == 4:
            h.update_S_POP(); o.update_S_POP()   # the inbreeding coefficients travel in the selfing-rate slots
            _same(h, o, ["self_rates", "state"] + pos, (it, "F"))
        h.update_ZQ(0); o.update_ZQ(0)
        _same(h, o, ["z", "qq", "qqnum"] + pos, (it, "ZQ"))
        h.update_alpha(); o.update_alpha()
        _same(h, o, ["alpha"] + pos, (it, "A"))
        h.cal_lkh(); o.cal_lkh()
        _same(h, o, ["indvlkh", "totallkh"], (it, "L"))
        o.lib.orc_iter_advance(o.h)
        h.lib.isg_iter_advance(h.h)
    assert o.error() == 0
    h.close()


@pytest.mark.parametrize("sched", [capi.SCHED_REPLAY, capi.SCHED_KEYED])
@pytest.mark.parametrize("case", [(50, 100, 3, 0.0, 2), (40, 1300, 6, 0.08, 3), (12, 70, 20, 0.3, 4)])
def test_mode0_every_sweep_bit_exact_vs_canonical_oracle(case, sched):
    """mode 0 (-v 0, no admixture): update_P / update_Z / cal_lkh; zz[i] travels in the generation slots"""
    N, L, K, miss, nall = case
    geno, an, mi = synth.code_diploid(synth.raw_alleles(N, L, K, 2, nall, miss, 17))
    h, o, initd = _pair(geno, an, mi, K, sched, 0)
    pos = ["seeds"] if sched == capi.SCHED_REPLAY else []
    h.chain_init(initd)
    o.chain_init(initd)
    _same(h, o, ["z", "generation"] + pos, "init")
    import ctypes as C
    o.lib.orc_iter_advance.argtypes = [C.c_void_p]
    h.lib.isg_iter_advance.argtypes = [C.c_void_p]
    for it in range(3):
        h.update_P(); o.update_P()
        _same(h, o, ["count_alleles", "freq"] + pos, (it, "P"))
        h.update_Z(0); o.update_Z(0)
        _same(h, o, ["z", "generation"] + pos, (it, "Z"))
        h.cal_lkh(); o.cal_lkh()
        _same(h, o, ["indvlkh", "totallkh"], (it, "L"))
        o.lib.orc_iter_advance(o.h)
        h.lib.isg_iter_advance(h.h)
    for it in range(2):
        h.iteration(); o.iteration()
        _same(h, o, ["z", "generation", "freq", "indvlkh", "totallkh"] + pos, ("iteration", it))
    assert o.error() == 0
    h.close()


@pytest.mark.parametrize("name", [n for n in sorted(gu.CASES) if gu.case_args(n)["mode"] == 0])
def test_mode0_replay_schedule_reproduces_reference_trajectory(name):
    c = gu.case_args(name)
    geno, an, mi = gu.case_data(name)
    lines = gu.parse(os.path.join(gu.GOLDEN, name + ".golden"))
    K, N = c["K"], geno.shape[0]
    h = capi.HipChain(geno, an, mi, K, mode=0, rng_sched=capi.SCHED_REPLAY)
    h.setseeds(*c["seeds"])
    initd = np.array([[h.ran1() for _ in range(K)] for _ in range(c["c"])], dtype=np.float32)
    it = iter(lines)
    cur = next(it)
    while not cur.startswith("init "):
        cur = next(it)
    assert gu.fields(cur)["seeds"] == h.seeds()

    def close(a, b, tol=1e-9):
        a, b = np.atleast_1d(np.asarray(a, dtype=float)), np.atleast_1d(np.asarray(b, dtype=float))
        return a.shape == b.shape and bool(np.all((a == b) | (np.abs(a - b) <= tol * np.abs(b))))
    convg = []
    for chn in range(c["c"]):
        h.chain_init(initd[chn])
        f = gu.fields(next(it))
        assert orc.fnv_i32(h.generation()) == f["hzz"] and f["seeds"] == h.seeds()
        zc, nstore = np.zeros((N, K), dtype=np.int64), 0
        for step in range(c["u"]):
            h.update_P()
            f = gu.fields(next(it))
            assert _counts_hash(h.count_alleles(), an) == f["hcnt"] and f["seeds"] == h.seeds(), (step, "P")
            h.update_Z(0)
            f = gu.fields(next(it))
            assert orc.fnv_i32(h.generation()) == f["hzz"] and f["seeds"] == h.seeds(), (step, "Z")
            h.cal_lkh()
            f = gu.fields(next(it))
            assert close(h.totallkh(), float.fromhex(f["totallkh"])), (step, "L")
            if step >= c["b"] and (step + 1 - c["b"]) % c["t"] == 0:
                zc[np.arange(N), h.generation()] += 1
                if nstore < c["r"]:
                    convg.append(h.totallkh())
                nstore += 1
        f = gu.fields(next(it))  # chain n done
        assert f["seeds"] == h.seeds()
        next(it)                 # chain steps= ...
        next(it)                 # chain indvlkh
        for i in range(N):
            assert [int(x) for x in next(it).split()[3:]] == list(zc[i])   # posterior assignment counts (CHAIN.z)
    line = next(it)
    assert line.startswith("convg") and close(convg, gu.floats(line))
    h.close()


EDGE_CASES = [
    # N, L, K, alleles, missing, note
    (40, 60, 6, 12, 0.05),     # microsatellite-like: count tile variant <256 lanes, 1 locus per lane>
    (20, 30, 20, 15, 0.05),    # Amax*K = 300: count tile variant <64, 1>; K > 16 kernel variant
    (12, 20, 32, 30, 0.0),     # Amax*K = 960: global-atomic count fallback; K = 32
    (9, 40, 1, 2, 0.1),        # a single cluster
    (2, 3, 2, 2, 0.0),         # two individuals, three loci
    (6, 70, 3, 2, 0.5),        # half of everything missing
]


@pytest.mark.parametrize("sched", [capi.SCHED_REPLAY, capi.SCHED_KEYED])
@pytest.mark.parametrize("case", EDGE_CASES)
def test_edge_shapes_bit_exact_vs_canonical_oracle(case, sched):
    N, L, K, nall, miss = case
    raw = synth.raw_alleles(N, L, max(K, 2), 2, nall, miss, 11)
    if N >= 6:
        raw[2] = synth.MISSING          # an individual without any usable locus
    if N == 2:
        raw[0, :, 0], raw[0, :, 1], raw[1, :, 0], raw[1, :, 1] = 1, 2, 2, 2   # keep every locus polymorphic
    geno, an, mi = synth.code_diploid(raw)
    h, o, initd = _pair(geno, an, mi, K, sched)
    assert h.keyed_layout() == o.keyed_layout()
    h.chain_init(initd)
    o.chain_init(initd)
    for it in range(3):
        h.iteration()
        o.iteration()
        _same(h, o, ["z", "qq", "qqnum", "generation", "alpha", "self_rates", "freq", "count_alleles", "indvlkh", "totallkh"], it)
        if sched == capi.SCHED_REPLAY:
            assert h.seeds() == o.seeds()
    h.close()


# replay update_ZQ kernel variants: the default picks k_zq_pipe (K <= 8, one locus per lane, Lp < 32768), k_zq_spec or
# k_zq_coop; the environment switches force the other ones; all of them must give the oracle's state
_S0 = {"INSTRUCT_ZQ_SPEC_RESOLVE": "0"}   # without the interval resolver (isg_spec_hip.inc), which is tried first by default: the block resolver
_R0 = dict(_S0, INSTRUCT_ZQ_RESOLVE="0")   # ... and without that: the chain kernels
_P0 = dict(_S0, INSTRUCT_ZQ_RESOLVE_PERSIST="0")   # one launch per block instead of k_zq_blocks (all blocks in one launch)
@pytest.mark.parametrize("env", [{}, _S0, dict(_S0, INSTRUCT_ZQ_RESOLVE_UNITS="16"), dict(_S0, INSTRUCT_ZQ_RESOLVE_A="0.6"), dict(_S0, INSTRUCT_ZQ_RESOLVE_A="12"),
                                 _P0, dict(_P0, INSTRUCT_ZQ_RESOLVE_UNITS="16"), dict(_P0, INSTRUCT_ZQ_RESOLVE_A="0.6"), dict(_P0, INSTRUCT_ZQ_RESOLVE_A="12"),
                                 _R0, dict(_R0, INSTRUCT_ZQ_PIPE="0"),
                                 dict(_R0, INSTRUCT_ZQ_PIPE_XCD="0"), dict(_R0, INSTRUCT_ZQ_SPEC="0"), dict(_R0, INSTRUCT_ZQ_XCD="1"),
                                 dict(_R0, INSTRUCT_ZQ_SPEC="0", INSTRUCT_ZQ_XCD="1"), dict(_R0, INSTRUCT_ZQ_COOP="0"),
                                 {"INSTRUCT_P_DEVICE": "0"}, {"INSTRUCT_LL_INT": "0"}, {"INSTRUCT_LL_TABLES": "0"}])   # (likelihood terms: double tables / evaluated directly)
@pytest.mark.parametrize("case", [(24, 700, 5, 0.05, 2), (6, 40000, 3, 0.02, 2), (8, 33000, 9, 0.0, 3),
                                  (10, 20000, 8, 0.03, 3), (16, 3000, 2, 0.1, 2)])
def test_replay_zq_kernel_variants_bit_exact_vs_canonical_oracle(case, env, monkeypatch):
    """L > 32768: more loci than 128 workgroups x 256 lanes -> several passes per individual in the cooperative kernels.
    INSTRUCT_ZQ_RESOLVE_UNITS=16: blocks of a few individuals; INSTRUCT_ZQ_RESOLVE_A=0.6: windows so narrow that most
    blocks end early on a miss; =12: windows wider than 64 candidates (the walk's second register set)"""
    N, L, K, miss, nall = case
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    geno, an, mi = synth.code_diploid(synth.raw_alleles(N, L, K, 2, nall, miss, 23))
    h, o, initd = _pair(geno, an, mi, K, capi.SCHED_REPLAY)
    h.chain_init(initd)
    o.chain_init(initd)
    _same(h, o, ["z", "qq", "qqnum", "seeds"], "init")
    for it in range(2):
        h.iteration()
        o.iteration()
        _same(h, o, ["z", "qq", "qqnum", "generation", "alpha", "self_rates", "freq", "indvlkh", "totallkh", "seeds"], it)
    h.close()


@pytest.mark.parametrize("persist", ["1", "0"])
@pytest.mark.parametrize("K", [1, 2, 3, 4, 6, 7, 8])
def test_replay_block_resolver_every_K_bit_exact(K, persist, monkeypatch):
    """The block kernels take K as a template constant (K = 1 shares K = 2's instance): every K the resolver handles, with all
    blocks in one launch and with one launch per block; three iterations against the canonical oracle, and the sweeps really
    went through the resolver."""
    monkeypatch.setenv("INSTRUCT_ZQ_RESOLVE_PERSIST", persist)
    monkeypatch.setenv("INSTRUCT_ZQ_SPEC_RESOLVE", "0")   # (the interval resolver would take these sweeps first)
    geno, an, mi = synth.code_diploid(synth.raw_alleles(150, 1300, max(K, 2), 2, 3, 0.04, 77 + K))
    h, o, initd = _pair(geno, an, mi, K, capi.SCHED_REPLAY)
    h.chain_init(initd)
    o.chain_init(initd)
    for it in range(3):
        h.iteration()
        o.iteration()
        _same(h, o, ["z", "qq", "qqnum", "generation", "alpha", "self_rates", "freq", "indvlkh", "totallkh", "seeds"], it)
    st = h.zq_resolve_stats()
    assert h.zq_fallbacks() == 0 and st["blocks"] >= 1 and (st["launches"] == 1) == (persist == "1")
    h.close()


@pytest.mark.parametrize("case", [(24, 700, 5, 0.05, 2), (8, 33000, 9, 0.0, 3)])
def test_replay_zq_aborted_cooperative_sweep_is_redone_bit_exact(case, monkeypatch):
    """INSTRUCT_ZQ_TEST_ABORT=2: the second cooperative update_ZQ sweep (= the first iteration's; the first is chain_init's)
    is treated as aborted AFTER it ran, i.e. with Z, qq and qqnum overwritten -- what a timed-out hand-off on a shared GPU
    leaves behind.  qq is restored from the copy taken before the launch, the single-workgroup kernel redoes the sweep from
    the same stream position: state and stream position equal the oracle's, and the chain carries on."""
    N, L, K, miss, nall = case
    monkeypatch.setenv("INSTRUCT_ZQ_TEST_ABORT", "2")
    monkeypatch.setenv("INSTRUCT_ZQ_SPEC_RESOLVE", "0")   # (the cooperative sweeps are what this test is about)
    geno, an, mi = synth.code_diploid(synth.raw_alleles(N, L, K, 2, nall, miss, 23))
    h, o, initd = _pair(geno, an, mi, K, capi.SCHED_REPLAY)
    h.chain_init(initd)
    o.chain_init(initd)
    assert h.zq_fallbacks() == 0
    for it in range(3):
        h.iteration()
        o.iteration()
        _same(h, o, ["z", "qq", "qqnum", "generation", "alpha", "self_rates", "freq", "indvlkh", "totallkh", "seeds"], it)
        assert h.zq_fallbacks() == 1
    h.close()


def test_selfing_rate_out_of_range_is_reported_like_the_reference():
    """dt_stat (mcmc.c:1524-1546) prints 'ERROR: The value of selfing rate or inbreeding coefficient %f is beyond [0,1]!' and
    exits when qq . S leaves [0,1]; update_G on the device raises the same message through the ABI instead of sampling on."""
    geno, an, mi = synth.code_diploid(synth.raw_alleles(30, 80, 3, 2, 2, 0.0, 3))
    for sched in (capi.SCHED_REPLAY, capi.SCHED_KEYED):
        h = capi.HipChain(geno, an, mi, 3, rng_sched=sched)
        h.setseeds(13, 4, 1972)
        h.chain_init(np.array([h.ran1() for _ in range(3)], dtype=np.float32))
        h.iteration()
        h.set_self_rates(np.array([1.5, 1.5, 1.5]))
        with pytest.raises(capi.IsgError, match=r"selfing rate or inbreeding coefficient 1\.500000 is beyond \[0,1\]"):
            h.update_G()
            h.cal_lkh()
            h.totallkh()   # keyed schedule: the flag comes back with the next likelihood download
        h.close()


@pytest.mark.parametrize("sched", [capi.SCHED_REPLAY, capi.SCHED_KEYED])
@pytest.mark.parametrize("mode", [2, 4])
def test_long_run_stays_bit_exact(mode, sched):
    """150 iterations at N=400, L=900, K=5 (several workgroups per individual): the rare paths of the samplers
    (rejected gamma attempts beyond the table, ambiguous Z draws redone in double, retries inside an attempt) all
    occur; the state is compared with the canonical oracle every 10 iterations."""
    geno, an, mi = synth.code_diploid(synth.raw_alleles(400, 900, 5, 2, 3, 0.03, 41))
    h, o, initd = _pair(geno, an, mi, 5, sched, mode)
    h.chain_init(initd)
    o.chain_init(initd)
    names = ["z", "qq", "qqnum", "alpha", "self_rates", "freq", "indvlkh", "totallkh"] + (["generation"] if mode == 2 else [])
    for blk in range(15):
        h.run(10)
        for _ in range(10):
            o.iteration()
        _same(h, o, names + (["seeds"] if sched == capi.SCHED_REPLAY else []), blk)
    assert o.error() == 0
    h.close()


@pytest.mark.parametrize("spec", ["1", "0"])
def test_config2_twenty_iterations_bit_exact(spec, monkeypatch):
    """BASELINE config 2 (N=2000, L=1000, K=5, mode 2), replay schedule: Z, allele counts, generations, qq and the
    stream position after each of 20 iterations against the canonical oracle (SURVEY 8d parity gate); with update_ZQ's start
    positions resolved from shape intervals (the default) and by the block resolver."""
    monkeypatch.setenv("INSTRUCT_ZQ_SPEC_RESOLVE", spec)
    geno, an, mi = synth.make_diploid(2000, 1000, 5)
    h, o, initd = _pair(geno, an, mi, 5, capi.SCHED_REPLAY, 2)
    h.chain_init(initd)
    o.chain_init(initd)
    for it in range(20):
        h.iteration()
        o.iteration()
        _same(h, o, ["z", "count_alleles", "generation", "qq", "self_rates", "alpha", "totallkh", "seeds"], it)
    # the sweeps really went through the resolution of the start positions (not the chain kernels behind it)
    st, sp, pd = h.zq_resolve_stats(), h.zq_spec_stats(), h.p_device_stats()
    assert h.zq_fallbacks() == 0 and pd["device_sweeps"] == 20 and pd["host_sweeps"] == 0, pd
    if spec == "1":
        assert sp["settled"] >= 18, sp
    else:
        # (launches: 1 = k_zq_blocks, all blocks in one launch; one launch per block when its workgroups cannot all be resident)
        assert sp["tried"] == 0 and st["blocks"] >= 2000 // 64 and (st["launches"] == 1 or st["launches"] > st["blocks"]) and 1 <= st["D"] <= 64
    h.close()


@pytest.mark.parametrize("spec", ["1", "0"])
def test_config3_two_iterations_bit_exact(full_size, spec, monkeypatch):
    """BASELINE config 3 (N=10000, L=5000, K=5), replay schedule, against the canonical oracle at full size
    (the oracle needs ~10 s per iteration here, hence two)."""
    monkeypatch.setenv("INSTRUCT_ZQ_SPEC_RESOLVE", spec)
    geno, an, mi = full_size
    h, o, initd = _pair(geno, an, mi, 5, capi.SCHED_REPLAY, 2)
    h.chain_init(initd)
    o.chain_init(initd)
    for it in range(2):
        h.iteration()
        o.iteration()
        _same(h, o, ["z", "count_alleles", "generation", "qq", "self_rates", "alpha", "totallkh", "seeds"], it)
    assert h.zq_fallbacks() == 0 and (h.zq_spec_stats()["settled"] == 2 if spec == "1" else h.zq_resolve_stats()["blocks"] > 100)
    assert h.p_device_stats()["device_sweeps"] == 2
    h.close()


def test_checkpoint_and_resume_continue_the_same_chain():
    """The sampler state at an iteration boundary is (z, qq, generation, selfing rates, alpha, stream position):
    a new context restored from it through the setters continues bit-identically (the reference cannot resume)."""
    geno, an, mi = synth.code_diploid(synth.raw_alleles(40, 300, 4, 2, 3, 0.05, 31))
    K = 4
    ref = capi.HipChain(geno, an, mi, K)
    ref.setseeds(13, 4, 1972)
    initd = np.array([ref.ran1() for _ in range(K)], dtype=np.float32)
    ref.chain_init(initd)
    ref.run(7)
    a = capi.HipChain(geno, an, mi, K)
    a.setseeds(13, 4, 1972)
    [a.ran1() for _ in range(K)]
    a.chain_init(initd)
    a.run(3)
    snap = dict(z=a.z(), qq=a.qq(), gen=a.generation(), S=a.self_rates(), alpha=a.alpha(), seeds=a.seeds())
    a.close()
    b = capi.HipChain(geno, an, mi, K)
    b.set_z(snap["z"]); b.set_qq(snap["qq"]); b.set_generation(snap["gen"]); b.set_self_rates(snap["S"]); b.set_alpha(snap["alpha"])
    b.setseeds(*snap["seeds"])
    b.run(4)
    for name in ("z", "qq", "qqnum", "generation", "self_rates", "freq", "indvlkh"):
        assert np.array_equal(getattr(b, name)(), getattr(ref, name)()), name
    assert b.alpha() == ref.alpha() and b.totallkh() == ref.totallkh() and b.seeds() == ref.seeds()
    ref.close(); b.close()


def _counts_hash(cnt, an):
    K, L, A = cnt.shape
    mask = np.arange(A)[None, :] < an[:, None]
    return orc.fnv_i32(cnt[:, mask])


def _runmean(m, x, step):
    """store_chn's multiplicative running mean (mcmc.c:1327-1332), vectorised."""
    return np.where(m != 0, m * ((step + x / np.where(m != 0, m, 1)) / (1 + step)), x / (1 + step))


@pytest.mark.parametrize("name", [n for n in sorted(gu.CASES) if gu.case_args(n)["mode"] != 0])
def test_replay_schedule_reproduces_reference_trajectory(name):
    """Golden trajectories come from the reference's own sweeps (oracle/ref_dump.c)."""
    c = gu.case_args(name)
    geno, an, mi = gu.case_data(name)
    lines = gu.parse(os.path.join(gu.GOLDEN, name + ".golden"))
    hdr = gu.fields([l for l in lines if l.startswith("data ")][0])
    assert orc.fnv_i32(geno) == hdr["hgeno"]  # same coded input as the reference reader produced
    K, N = c["K"], geno.shape[0]
    h = capi.HipChain(geno, an, mi, K, mode=c["mode"], type_freq=c["y"], back_refl=c["e"], rng_sched=capi.SCHED_REPLAY)
    h.setseeds(*c["seeds"])
    initd = np.array([[h.ran1() for _ in range(K)] for _ in range(c["c"])], dtype=np.float32)  # read_init, initial.c:56-61
    it = iter(lines)
    cur = next(it)
    while not cur.startswith("init"):
        cur = next(it)
    while cur.startswith("initd"):
        k = int(cur.split()[1])
        assert gu.floats(cur) == [float(v) for v in initd[k]]
        cur = next(it)
    assert gu.fields(cur)["seeds"] == h.seeds()

    def close(a, b, tol=1e-9):
        a, b = np.atleast_1d(np.asarray(a, dtype=float)), np.atleast_1d(np.asarray(b, dtype=float))
        return a.shape == b.shape and bool(np.all((a == b) | (np.abs(a - b) <= tol * np.abs(b))))

    convg = []
    for chn in range(c["c"]):
        h.chain_init(initd[chn])
        f = gu.fields(next(it))           # chain n init alpha= seeds=
        assert close(h.alpha(), float.fromhex(f["alpha"]))
        if c["mode"] in (2, 3):
            f = gu.fields(next(it))       # geninit
            assert orc.fnv_i32(h.generation()) == f["hgen"]
        if c["mode"] == 5:
            f = gu.fields(next(it))       # initial coefficients: uniforms, exact
            assert orc.fnv_f64(h.self_rates()) == f["hS"]
        f = gu.fields(next(it))           # zqinit
        assert orc.fnv_i32(h.z()) == f["hz"] and f["seeds"] == h.seeds()
        res, cnt_step = None, 0
        for step in range(c["u"]):
            h.update_P()
            f = gu.fields(next(it))
            assert _counts_hash(h.count_alleles(), an) == f["hcnt"] and f["seeds"] == h.seeds(), (step, "P")
            if c["mode"] == 2:
                h.update_S_POP()
                line = next(it)
                assert close(h.self_rates(), gu.floats(line)[:K]) and gu.fields(line)["seeds"] == h.seeds(), (step, "S")
                if c["e"] == 0:
                    assert ["st%d" % s for s in h.state()] == [t for t in line.split() if t.startswith("st")]
                h.update_G()
                f = gu.fields(next(it))
                assert orc.fnv_i32(h.generation()) == f["hgen"] and f["seeds"] == h.seeds(), (step, "G")
            if c["mode"] == 3:
                h.update_S_IND()
                line = next(it)
                assert close(h.self_rates()[:4], gu.floats(line)[:4]) and gu.fields(line)["seeds"] == h.seeds(), (step, "SI")
                h.update_G()
                f = gu.fields(next(it))
                assert orc.fnv_i32(h.generation()) == f["hgen"] and f["seeds"] == h.seeds(), (step, "G")
            if c["mode"] == 5:
                h.update_S_IND()   # update_F_IND
                line = next(it)
                assert close(h.self_rates()[:4], gu.floats(line)[:4]) and gu.fields(line)["seeds"] == h.seeds(), (step, "FI")
            if c["mode"] == 4:
                h.update_S_POP()   # update_inbreedcoff_POP: the coefficients are dumped on the S line
                line = next(it)
                assert close(h.self_rates(), gu.floats(line)[:K]) and gu.fields(line)["seeds"] == h.seeds(), (step, "F")
                if c["e"] == 0:
                    assert ["st%d" % s for s in h.state()] == [t for t in line.split() if t.startswith("st")]
            h.update_ZQ(0)
            f = gu.fields(next(it))
            assert orc.fnv_i32(h.z()) == f["hz"] and orc.fnv_f64(h.qqnum()) == f["hqqnum"] and f["seeds"] == h.seeds(), (step, "ZQ")
            h.update_alpha()
            f = gu.fields(next(it))
            assert close(h.alpha(), float.fromhex(f["alpha"])) and f["seeds"] == h.seeds(), (step, "A")
            h.cal_lkh()
            f = gu.fields(next(it))
            assert close(h.totallkh(), float.fromhex(f["totallkh"])), (step, "L")
            if c["detail"] > 0 and step % c["detail"] == 0:
                qq, fr = h.qq(), h.freq()
                for i in range(min(N, 4)):
                    assert close(qq[i], gu.floats(next(it)))
                for k in range(K):
                    for j in range(min(geno.shape[1], 4)):
                        assert close(fr[k, j, :an[j]], gu.floats(next(it)))
            # CHAIN bookkeeping as the driver does it (mcmc.c:218-226)
            if step == c["b"] - 1:
                res = {"n": 0, "totallkh": 1.0, "qq": np.ones((N, K)), "S": np.ones(N if c["mode"] in (3, 5) else K), "gen": np.ones(N), "indv": np.ones(N)}
            if step >= c["b"] and (step + 1 - c["b"]) % c["t"] == 0:
                s = res["n"]
                res["totallkh"] = float(_runmean(np.float64(res["totallkh"]), h.totallkh(), s))
                res["qq"] = _runmean(res["qq"], h.qq(), s)
                res["indv"] = _runmean(res["indv"], h.indvlkh(), s)
                if c["mode"] in (2, 3, 4, 5):
                    res["S"] = _runmean(res["S"], h.self_rates(), s)
                if c["mode"] in (2, 3):
                    res["gen"] = _runmean(res["gen"], h.generation().astype(float), s)
                res["n"] += 1
                if cnt_step < c["r"]:
                    convg.append(h.totallkh())
                cnt_step += 1
        f = gu.fields(next(it))  # chain n done seeds=
        assert f["seeds"] == h.seeds()
        # posterior means: north-star tolerance 1e-6 relative (we are ~1e-12)
        f = gu.fields(next(it))
        assert int(f["step"]) == res["n"] and close(res["totallkh"], float.fromhex(f["totallkh"]), 1e-6)
        assert close(res["indv"], gu.floats(next(it)), 1e-6)
        if c["mode"] in (4, 5):
            assert close(res["S"], gu.floats(next(it)), 1e-6)
            next(it)
        if c["mode"] in (2, 3):
            assert close(res["S"], gu.floats(next(it)), 1e-6)
            next(it)
            assert close(res["gen"], gu.floats(next(it)), 1e-6)
            next(it)
        for i in range(N):
            assert close(res["qq"][i], gu.floats(next(it))[:K], 1e-6)
        if c["pf"] == 1:
            next(it)
    line = next(it)
    assert line.startswith("convg") and close(convg, gu.floats(line))
    if c["c"] > 1:
        gr = gu.floats(next(it))[0]
        assert close(capi.gelman_rubin(np.array(convg), c["c"], c["r"]), gr, 1e-6)
    h.close()


def test_dropin_cli_output_equals_reference_cli_output(tmp_path):
    """The reference driver (InStruct.c, data_interface.c, result_analysis.c ... compiled from the
    reference sources into oracle/_ref) linked with instruct_amd/host/mcmc_hip.c instead of mcmc.c
    writes the same result file as the pure reference binary did (tests/golden/c1_cli_output.txt)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "InStruct_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/InStruct_hip not built (needs the reference objects; built in the dev container)")
    out = tmp_path / "out.txt"
    cmd = [exe, "-d", os.path.join(gu.GOLDEN, "c1.txt"), "-o", str(out), "-K", "3", "-L", "100", "-N", "50", "-p", "2",
           "-u", "200", "-b", "100", "-t", "10", "-c", "2", "-v", "2", "-g", "1", "-r", "5", "-j", "5",
           "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "0", "-pf", "1"]
    log = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert log.returncode == 0 and b"THE JOB IS SUCCESSFULLY FINISHED" in log.stdout, log.stdout[-2000:]

    def body(path):
        keep = []
        for l in open(path, "rb").read().split(b"\n"):
            if l.strip().startswith((b"Data File:", b"Output File:")) or b"InStruct" in l and b"-d" in l:
                continue
            keep.append(l)
        return keep
    assert body(str(out)) == body(os.path.join(gu.GOLDEN, "c1_cli_output.txt"))


def test_dropin_cli_output_equals_reference_cli_output_ploidy4(tmp_path):
    """Same for `-p 4 -ap 1`: mcmc_updating routes to the MI355X ploidy-4 chain instead of poly_geno.c's driver
    (tests/golden/t1_cli_output.txt was written by the pure reference binary)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "InStruct_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/InStruct_hip not built (needs the reference objects; built in the dev container)")
    out = tmp_path / "out.txt"
    cmd = [exe, "-d", os.path.join(gu.GOLDEN, "t1.txt"), "-o", str(out)] + gu.make_golden.TETRA_CLI
    log = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert log.returncode == 0 and b"THE JOB IS SUCCESSFULLY FINISHED" in log.stdout, log.stdout[-2000:]

    def body(path):
        return [l for l in open(path, "rb").read().split(b"\n")
                if not (l.strip().startswith((b"Data File:", b"Output File:")) or b"InStruct" in l and b"-d" in l)]
    assert body(str(out)) == body(os.path.join(gu.GOLDEN, "t1_cli_output.txt"))


def test_dropin_cli_output_equals_reference_cli_output_allotetraploid(tmp_path):
    """`-p 4 -ap 0`: the drop-in runs the allotetraploid chain on the device (no reference code left in that path:
    tests/golden/ta1_cli_output.txt was written by the pure reference binary)"""
    exe = os.path.join(ROOT, "oracle", "_ref", "InStruct_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/InStruct_hip not built (needs the reference objects; built in the dev container)")
    out = tmp_path / "out.txt"
    cmd = [exe, "-d", os.path.join(gu.GOLDEN, "ta1.txt"), "-o", str(out)] + gu.make_golden.ALLO_CLI
    log = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert log.returncode == 0 and b"THE JOB IS SUCCESSFULLY FINISHED" in log.stdout, log.stdout[-2000:]

    def body(path):
        return [l for l in open(path, "rb").read().split(b"\n")
                if not (l.strip().startswith((b"Data File:", b"Output File:")) or b"InStruct" in l and b"-d" in l)]
    assert body(str(out)) == body(os.path.join(gu.GOLDEN, "ta1_cli_output.txt"))


@pytest.mark.parametrize("which", ["mode0", "mode3", "mode4", "mode5"])
def test_dropin_cli_output_equals_reference_cli_output_other_modes(which, tmp_path):
    """`-v 3 -f 0` (one selfing rate per individual, uniform prior) and `-v 4 -e 0` (population inbreeding
    coefficients, adaptive independence proposals) through the drop-in"""
    exe = os.path.join(ROOT, "oracle", "_ref", "InStruct_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/InStruct_hip not built (needs the reference objects; built in the dev container)")
    out = tmp_path / "m.txt"
    cmd = [exe, "-d", os.path.join(gu.GOLDEN, "c1.txt"), "-o", str(out)] + {"mode0": gu.make_golden.MODE0_CLI, "mode3": gu.make_golden.MODE3_CLI, "mode4": gu.make_golden.MODE4_CLI, "mode5": gu.make_golden.MODE5_CLI}[which]
    log = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert log.returncode == 0 and b"THE JOB IS SUCCESSFULLY FINISHED" in log.stdout, log.stdout[-2000:]

    def body(path):
        return [l for l in open(path, "rb").read().split(b"\n")
                if not (l.strip().startswith((b"Data File:", b"Output File:")) or b"InStruct" in l and b"-d" in l)]
    assert body(str(out)) == body(os.path.join(gu.GOLDEN, "c1_%s_cli_output.txt" % which))


def test_dropin_k_scan_output_equals_reference(tmp_path):
    """`-ik 1 -kv 2 3`: the driver runs every chain for K = 2 and K = 3 and keeps the K with the smallest DIC
    (InStruct.c:536-601); the drop-in re-creates its device context when K changes."""
    exe = os.path.join(ROOT, "oracle", "_ref", "InStruct_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/InStruct_hip not built (needs the reference objects; built in the dev container)")
    out = tmp_path / "k.txt"
    cmd = [exe, "-d", os.path.join(gu.GOLDEN, "c1.txt"), "-o", str(out)] + gu.make_golden.KSCAN_CLI
    log = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert log.returncode == 0 and b"THE JOB IS SUCCESSFULLY FINISHED" in log.stdout, log.stdout[-2000:]

    def body(path):
        return [l for l in open(path, "rb").read().split(b"\n")
                if not (l.strip().startswith((b"Data File:", b"Output File:")) or b"InStruct" in l and b"-d" in l)]
    assert body(str(out)) == body(os.path.join(gu.GOLDEN, "c1_kscan_output.txt"))


def test_reader_and_sampler_replaced_cli_output_equals_reference(tmp_path):
    """oracle/_ref/InStruct_full = the reference driver around BOTH of this repository's objects (streaming reader
    + MI355X sampler): same result files as the pure reference binary, diploid and ploidy 4."""
    exe = os.path.join(ROOT, "oracle", "_ref", "InStruct_full")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/InStruct_full not built (needs the reference objects; built in the dev container)")

    def body(path):
        return [l for l in open(path, "rb").read().split(b"\n")
                if not (l.strip().startswith((b"Data File:", b"Output File:")) or b"InStruct" in l and b"-d" in l)]
    out = tmp_path / "d.txt"
    cmd = [exe, "-d", os.path.join(gu.GOLDEN, "c1.txt"), "-o", str(out), "-K", "3", "-L", "100", "-N", "50", "-p", "2",
           "-u", "200", "-b", "100", "-t", "10", "-c", "2", "-v", "2", "-g", "1", "-r", "5", "-j", "5",
           "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "0", "-pf", "1"]
    log = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert log.returncode == 0 and b"THE JOB IS SUCCESSFULLY FINISHED" in log.stdout, log.stdout[-2000:]
    assert body(str(out)) == body(os.path.join(gu.GOLDEN, "c1_cli_output.txt"))
    out4 = tmp_path / "t.txt"
    cmd = [exe, "-d", os.path.join(gu.GOLDEN, "t1.txt"), "-o", str(out4)] + gu.make_golden.TETRA_CLI
    log = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert log.returncode == 0 and b"THE JOB IS SUCCESSFULLY FINISHED" in log.stdout, log.stdout[-2000:]
    assert body(str(out4)) == body(os.path.join(gu.GOLDEN, "t1_cli_output.txt"))


def _stdout_equal(got, want):
    """Program stdout, line by line.  print_info prints doubles at %f (mcmc.c:1275-1313): the device's doubles agree
    with the reference's to ~1e-12 relative, so a printed 6th decimal may round the other way -- numbers are compared
    within 2e-6, everything else byte for byte."""
    import re
    num = re.compile(rb"-?\d+\.\d{6}")
    g, w = got.split(b"\n"), want.split(b"\n")
    assert len(g) == len(w), (len(g), len(w))
    for a, b in zip(g, w):
        if a == b:
            continue
        assert num.sub(b"#", a) == num.sub(b"#", b), (a[:200], b[:200])
        for x, y in zip(num.findall(a), num.findall(b)):
            assert abs(float(x) - float(y)) <= 2e-6, (a[:200], b[:200])


@pytest.mark.parametrize("name", sorted(gu.make_golden.STDOUT_CLI))
def test_dropin_print_info_and_empty_cluster_restart_equal_reference(name, tmp_path):
    """`-pi 1` (print_info, mcmc.c:1267-1316) for modes 2, 3, 4 (-e 0: states printed), 5 and ploidy 4, and -- case ec1
    -- a chain that trips check_empty_cluster (mcmc.c:1944-1974) at its 5th stored step, is discarded and re-run with
    the stream continuing (InStruct.c:185-190): the drop-in's STDOUT and result file equal the pure reference binary's
    (tests/golden/<name>_cli_stdout.txt, _cli_output.txt)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "InStruct_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/InStruct_hip not built (needs the reference objects; built in the dev container)")
    data, cli = gu.make_golden.STDOUT_CLI[name]
    out = tmp_path / "o.txt"
    cmd = [exe, "-d", os.path.join(gu.GOLDEN, data), "-o", str(out)] + cli
    log = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
    assert log.returncode == 0 and b"THE JOB IS SUCCESSFULLY FINISHED" in log.stdout, log.stdout[-2000:]
    want = open(os.path.join(gu.GOLDEN, name + "_cli_stdout.txt"), "rb").read()
    if name == "ec1":
        assert want.count(b"has an empty cluster, thus discarded!") == 1 and want.count(b"Chain#2 Starts:") == 2
    _stdout_equal(log.stdout, want)

    def body(path):
        return [l for l in open(path, "rb").read().split(b"\n")
                if not (l.strip().startswith((b"Data File:", b"Output File:")) or b"InStruct" in l and b"-d" in l)]
    assert body(str(out)) == body(os.path.join(gu.GOLDEN, name + "_cli_output.txt"))


@pytest.fixture(scope="module")
def full_size():
    geno, an, mi = synth.make_diploid(10000, 5000, 5)
    return geno, an, mi


def test_full_size_properties_config3(full_size):
    """N=10000, L=5000, K=5 (BASELINE config 3): properties that hold at any size."""
    geno, an, mi = full_size
    N, L, _ = geno.shape
    K = 5
    hashes = {}
    for sched in (capi.SCHED_REPLAY, capi.SCHED_KEYED):
        for rep in range(2 if sched == capi.SCHED_KEYED else 1):
            h = capi.HipChain(geno, an, mi, K, rng_sched=sched)
            h.setseeds(13, 4, 1972)
            initd = np.array([h.ran1() for _ in range(K)], dtype=np.float32)
            h.chain_init(initd)
            h.run(2)
            z = h.z()
            valid = mi == 0
            assert (z[valid] >= 0).all() and (z[valid] < K).all() and (z[~valid] == -1).all()
            # allele counts == bincount of (z, locus, allele) over valid copies; total = 2 * #valid loci
            cnt = h.count_alleles()
            idx = (z.astype(np.int64) * L + np.arange(L)[None, :, None]) * h.Amax + geno
            ref = np.bincount(idx[np.repeat(valid[:, :, None], 2, 2)], minlength=K * L * h.Amax).reshape(K, L, h.Amax)
            assert np.array_equal(cnt, ref) and cnt.sum() == 2 * valid.sum()
            # qqnum rows = histogram of the individual's Z; qq rows are probability vectors
            qn = h.qqnum()
            zz = np.where(valid[:, :, None], z, K).reshape(N, -1)
            assert np.array_equal(qn, np.stack([(zz == k).sum(1) for k in range(K)], 1).astype(float))
            assert np.allclose(h.qq().sum(1), 1.0, atol=1e-12) and (h.qq() > 0).all()
            fr = h.freq()
            assert np.allclose(fr.sum(2), 1.0, atol=1e-12)
            g = h.generation()
            assert g.min() >= 1 and g.max() <= 50
            assert np.isfinite(h.indvlkh()).all() and h.totallkh() < 0
            hashes.setdefault(sched, []).append((orc.fnv_i32(z), orc.fnv_f64(h.qq()), orc.fnv_i32(g), h.totallkh()))
            h.close()
    # keyed schedule: two runs with the same seeds are identical to the last bit (order-independent sums)
    assert hashes[capi.SCHED_KEYED][0] == hashes[capi.SCHED_KEYED][1]


def _runmean(m, x, n):
    """the reference's multiplicative running mean (mcmc.c:1327-1332), element-wise in IEEE double"""
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(m != 0, m * ((n + x / m) / (1 + n)), x / (1 + n))


@pytest.mark.parametrize("sched", [capi.SCHED_REPLAY, capi.SCHED_KEYED])
def test_store_chn_on_device_equals_the_host_running_means(sched):
    """isg_store_step keeps CHAIN.qq/qq2/indvlkh/gen/gen2/freq/freq2 on the device: same bits as store_chn's loop
    (mcmc.c:1320-1456) applied to the downloaded state"""
    N, L, K = 60, 130, 4
    geno, an, mi = synth.code_diploid(synth.raw_alleles(N, L, K, 2, 3, 0.05, 5))
    h = capi.HipChain(geno, an, mi, K, rng_sched=sched)
    h.setseeds(13, 4, 1972)
    h.chain_init(np.array([h.ran1() for _ in range(K)], dtype=np.float32))
    for _ in range(3):
        h.iteration()
    h.store_begin(with_freq=True)
    ref = {k: np.ones(s) for k, s in (("qq", (N, K)), ("qq2", (N, K)), ("indvlkh", (N,)), ("gen", (N,)), ("gen2", (N,)),
                                      ("freq", (K, L, h.Amax)), ("freq2", (K, L, h.Amax)))}
    for n in range(7):
        h.iteration()
        if n % 2:
            h.iteration()  # a thinning step in between
        h.store_step()
        qq, lk, g, f = h.qq(), h.indvlkh(), h.generation().astype(np.int64), h.freq()
        for key, x in (("qq", qq), ("qq2", qq * qq), ("indvlkh", lk), ("gen", g.astype(np.float64)), ("gen2", (g * g).astype(np.float64)),
                       ("freq", f), ("freq2", f * f)):
            ref[key] = _runmean(ref[key], x, n)
    got = h.store_fetch(("qq", "qq2", "indvlkh", "gen", "gen2", "freq", "freq2"))
    assert got["steps"] == 7
    for key in ref:
        a, b = got[key], ref[key]
        if key.startswith("freq"):  # only alleles that exist are part of CHAIN.freq (mcmc.c:1438-1441)
            mask = np.arange(h.Amax)[None, None, :] < np.asarray(an)[None, :, None]
            a, b = a[np.broadcast_to(mask, a.shape)], b[np.broadcast_to(mask, b.shape)]
        assert np.array_equal(a, b), key
    h.close()
