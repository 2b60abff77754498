"""GPU: ploidy 4 (autotetraploid) sweeps on the MI355X against
 (1) the canonical configuration of the oracle (oracle/orc_dump_poly ... 1 1): every dump line identical,
     i.e. genotypes, Z, counts, float tables, selfing rates, qq, likelihoods and seeds bit for bit;
 (2) the golden trajectories generated from the REAL reference (tests/golden/t*.golden): discrete state
     (imputed genotypes, Z, counts, MH states, seeds) identical, doubles within 1e-9 relative."""
import os
import subprocess

import numpy as np
import pytest

import golden_util as gu
import orc

pytestmark = pytest.mark.gpu
DUMP = os.path.join(orc.ORC_DIR, "orc_dump_poly")
POLY = gu.make_golden.POLY_CASES
ALLO = gu.make_golden.ALLO_CASES


# generated at test time (oracle run on the GPU box's host): more alleles (6: 126 genotypes per locus; 10: 715), more
# clusters, both proposal kinds
EXTRA = {
    "x_a4k5": (120, 200, 5, 4, 0.05, 3, 1, 1, 1, 1, 1, (21, 7, 1999)),
    "x_a6k2": (40, 30, 2, 6, 0.05, 4, 1, 1, 0, 1, 1, (22, 8, 2000)),
    "x_a5k12": (50, 70, 12, 5, 0.10, 3, 1, 1, 1, 1, 1, (23, 9, 2001)),
    "x_a3k20": (30, 700, 20, 3, 0.05, 2, 1, 1, 1, 1, 1, (24, 10, 2002)),  # several workgroups per individual, K > 16
    "x_a4k3": (25, 1500, 3, 4, 0.20, 2, 1, 1, 0, 1, 1, (25, 11, 2003)),
    "x_wide": (3, 66500, 3, 3, 0.05, 2, 1, 1, 1, 1, 1, (26, 12, 2004)),   # more loci than 128 workgroups x 512 lanes: several passes
    "x_a10k3": (60, 12, 3, 10, 0.05, 2, 1, 1, 1, 1, 1, (27, 13, 2005)),   # microsatellite-like: 10 alleles, 715 genotypes per locus
    "x_spec": (120, 2500, 4, 4, 0.05, 3, 1, 1, 1, 1, 1, (28, 14, 2006)),    # clusters of ~2400 draws: the interval resolver settles update_ZQ itself
    "x_zz_a16k3": (80, 8, 3, 16, 0.05, 2, 1, 1, 1, 1, 1, (29, 15, 2007)),   # the allele cap: 16 alleles, 3876 genotypes per locus
}


def extra_data(name):
    from instruct_amd import synth
    N, L, K, A, miss = EXTRA[name][:5]
    return synth.raw_alleles(N, L, K, 4, A, miss, 20260301 + sorted(EXTRA).index(name))


def hip_lines(name, cfg=None, raw=None, sched=0, fast_coder=False, allo=False):
    """Drives the C ABI sweep by sweep and formats the state as oracle/isg_oracle_poly.c's dump does."""
    from instruct_amd import capi, synth
    N, L, K, A, miss, u, b, t, e, r, j, seeds = cfg or (ALLO[name] if allo else POLY[name])
    coder = synth.code_tetraploid_fast if fast_coder else synth.code_tetraploid
    if raw is None:
        raw = gu.make_golden.allo_data_for(name) if allo else gu.make_golden.poly_data_for(name)
    obs, alleleid, allelenum = coder(raw)
    ch = capi.HipPolyChain(obs, alleleid, allelenum, K, back_refl=e, rng_sched=sched, allo=allo)
    ch.setseeds(*seeds)
    initd = np.array([np.float32(ch.ran1()) for _ in range(K)], dtype=np.float32)
    lines = []
    sd = lambda: " seeds=%d %d %d" % ch.seeds()
    # the reference draws alpha, imputes the genotypes, then update_ZQ(1): chain_init does the three
    ch.chain_init(initd)
    lines.append("chain zqinit hz=%s hqq=%s" % (orc.fnv_i32(ch.z()), orc.fnv_f64(ch.qq())) + sd())
    for step in range(u):
        ch.update_P()
        lines.append("it %d P hcnt=%s hfreq=%s" % (step, _hcnt(ch, allelenum), _hfreq(ch, allelenum)) +
                     (" hfreq2=%s" % _hfreq(ch, allelenum, ch.freq2()) if allo else "") + sd())
        lines.append("it %d X hexfreq=%s" % (step, orc.fnv_i32(ch.packed(ch.exfreq()).view(np.int32))))
        ch.update_S_POP()
        s = "it %d S" % step + "".join(" " + float(x).hex() for x in ch.self_rates())
        if e == 0:
            s += "".join(" st%d" % x for x in ch.state())
        lines.append(s + " hgenofreq=%s" % orc.fnv_i32(ch.packed(ch.genofreq()).view(np.int32)) + sd())
        ch.update_ZQ(0)
        lines.append("it %d ZQ hz=%s hqq=%s hqqnum=%s" % (step, orc.fnv_i32(ch.z()), orc.fnv_f64(ch.qq()), orc.fnv_f64(ch.qqnum())) + sd())
        ch.update_geno()
        lines.append("it %d GE hgeno=%s" % (step, orc.fnv_i32(ch.geno())) + sd())
        ch.cal_lkh()
        lines.append("it %d L totallkh=%s hindv=%s" % (step, float(ch.totallkh()).hex(), orc.fnv_f64(ch.indvlkh())))
        ch.lib.isg_iter_advance(ch.h)   # sweep-by-sweep drivers count the iterations themselves (keyed positions)
    ch.close()
    return lines


def _hcnt(ch, allelenum):
    c = ch.count_alleles()
    mask = np.arange(ch.Amax)[None, :] < allelenum[:, None]
    return orc.fnv_i32(np.ascontiguousarray(c[:, mask]))


def _hfreq(ch, allelenum, f=None):
    f = ch.freq() if f is None else f
    mask = (np.arange(ch.Amax)[None, :] < allelenum[:, None]) & (allelenum[:, None] > 1)
    return orc.fnv_f64(np.ascontiguousarray(f[:, mask]))


def _norm(line):
    """C's %a and Python's float.hex differ in trailing zeros / exponent sign: compare parsed values"""
    toks = []
    for t in line.split():
        k, eq, v = t.partition("=")
        cand = v if eq else t
        if cand.startswith(("0x", "-0x")) or cand in ("inf", "-inf", "nan"):
            toks.append((k if eq else "") + repr(float.fromhex(cand) if "x" in cand else float(cand)))
        else:
            toks.append(t)
    return " ".join(toks)


@pytest.fixture(scope="module", autouse=True)
def _build():
    orc.build()


@pytest.mark.parametrize("name", sorted(POLY))
def test_tetraploid_bit_identical_to_canonical_oracle(name, tmp_path):
    N, L, K, A, miss, u, b, t, e, r, j, seeds = POLY[name]
    out = str(tmp_path / (name + ".can"))
    args = [DUMP, os.path.join(gu.GOLDEN, name + ".txt"), out] + [str(x) for x in (K, N, L, u, b, t, e, r, j) + tuple(seeds)] + ["1", "1"]
    assert subprocess.call(args) == 0
    want = [l for l in gu.parse(out) if l.startswith("it ") or l.startswith("chain zqinit")]
    got = hip_lines(name)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert _norm(g) == _norm(w)


@pytest.mark.parametrize("name", ["t1", "x_a3k20", "x_a4k3", "xa_a3k9"])
def test_tetraploid_block_resolved_update_zq_gives_the_same_lines(name, monkeypatch):
    """INSTRUCT_ZQ_RESOLVE_P4=1: replay update_ZQ with the start positions resolved block-wise (k4_zq_block) and one
    parallel sweep at them -- same lines as the cooperative chain kernel (which the other tests pin to the oracle)"""
    allo = name.startswith("xa_")
    cfg = None if name in POLY else (ALLO_EXTRA[name] if allo else EXTRA[name])
    raw = None
    if cfg is not None:
        from instruct_amd import synth
        raw = synth.raw_alleles(cfg[0], cfg[1], cfg[2], 4, cfg[3], cfg[4], 20260401 + sorted(ALLO_EXTRA).index(name)) if allo else extra_data(name)
    want = hip_lines(name, cfg, raw, allo=allo)
    monkeypatch.setenv("INSTRUCT_ZQ_RESOLVE_P4", "1")
    monkeypatch.setenv("INSTRUCT_ZQ_SPEC_RESOLVE", "0")   # (the interval resolver would be tried first)
    assert hip_lines(name, cfg, raw, allo=allo) == want


def test_tetraploid_single_workgroup_zq_kernel_gives_the_same_lines(monkeypatch):
    """INSTRUCT_ZQ_COOP=0 selects the one-workgroup update_ZQ kernel (no uniform tape, no hand-offs)"""
    coop = hip_lines("t1")
    monkeypatch.setenv("INSTRUCT_ZQ_COOP", "0")
    monkeypatch.setenv("INSTRUCT_ZQ_SPEC_RESOLVE", "0")
    assert hip_lines("t1") == coop


def test_tetraploid_aborted_cooperative_sweep_is_redone_bit_exact(monkeypatch):
    """INSTRUCT_ZQ_TEST_ABORT=2: the first iteration's cooperative update_ZQ is treated as aborted after it ran; qq is
    restored and the single-workgroup kernel redoes the sweep -- every later dump line unchanged"""
    want = hip_lines("t1")
    monkeypatch.setenv("INSTRUCT_ZQ_TEST_ABORT", "2")
    monkeypatch.setenv("INSTRUCT_ZQ_SPEC_RESOLVE", "0")
    assert hip_lines("t1") == want


# ---------------------------------------------------------------------------------------------- allotetraploid, -ap 0
ALLO_EXTRA = {
    "xa_a4k5": (60, 120, 5, 4, 0.05, 3, 1, 1, 1, 1, 1, (31, 7, 1999)),
    "xa_a6k2": (30, 24, 2, 6, 0.05, 3, 1, 1, 0, 1, 1, (32, 8, 2000)),    # 6 alleles: 441 genotypes per locus
    "xa_a3k9": (20, 700, 9, 3, 0.10, 2, 1, 1, 1, 1, 1, (33, 9, 2001)),   # several workgroups per individual, K > 8
    "xa_a9k3": (50, 10, 3, 9, 0.05, 2, 1, 1, 1, 1, 1, (34, 10, 2002)),   # 9 alleles: 2673 genotypes per locus
    "xa_zz_a16k2": (50, 6, 2, 16, 0.05, 2, 1, 1, 1, 1, 1, (35, 11, 2003)),  # the allele cap: 16 alleles, 18496 genotypes per locus
}


def _allo_oracle(txt, out, cfg, canonical):
    N, L, K, A, miss, u, b, t, e, r, j, seeds = cfg
    args = [DUMP, txt, out] + [str(x) for x in (K, N, L, u, b, t, e, r, j) + tuple(seeds)] + (["1", "1"] if canonical else ["0", "0"]) + ["0", "1"]
    assert subprocess.call(args) == 0
    return [l for l in gu.parse(out) if l.startswith("it ") or l.startswith("chain zqinit")]


@pytest.mark.parametrize("name", sorted(ALLO))
def test_allotetraploid_bit_identical_to_canonical_oracle(name, tmp_path):
    """-p 4 -ap 0 on the device (update_P_allo, calc_exfreq_allo, allo_genfreq, choose_*_allo, two-subgenome likelihood):
    every dump line of the three golden cases equal to the canonical oracle's"""
    want = _allo_oracle(os.path.join(gu.GOLDEN, name + ".txt"), str(tmp_path / (name + ".can")), ALLO[name], True)
    got = hip_lines(name, allo=True)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert _norm(g) == _norm(w)


@pytest.mark.parametrize("name", sorted(ALLO_EXTRA))
def test_allotetraploid_generated_cases_bit_identical_to_canonical_oracle(name, tmp_path):
    from instruct_amd import synth
    cfg = ALLO_EXTRA[name]
    raw = synth.raw_alleles(cfg[0], cfg[1], cfg[2], 4, cfg[3], cfg[4], 20260401 + sorted(ALLO_EXTRA).index(name))
    txt = str(tmp_path / (name + ".txt"))
    synth.write_text_polyploid(txt, raw)
    want = _allo_oracle(txt, str(tmp_path / (name + ".can")), cfg, True)
    got = hip_lines(name, cfg, raw, allo=True)
    assert len(got) == len(want) == 1 + 6 * cfg[5]
    for g, w in zip(got, want):
        assert _norm(g) == _norm(w)


@pytest.mark.parametrize("name", sorted(ALLO))
def test_allotetraploid_matches_reference_golden(name):
    """against the trajectories the REAL reference produced (oracle/ref_dump_poly.c ... 0): discrete state and seeds
    identical after every sweep, doubles within 1e-9"""
    want = [l for l in gu.parse(os.path.join(gu.GOLDEN, name + ".golden")) if l.startswith("it ") or l.startswith("chain zqinit")]
    got = hip_lines(name, allo=True)
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


@pytest.mark.parametrize("name", sorted(ALLO) + sorted(ALLO_EXTRA))
def test_allotetraploid_keyed_schedule_bit_identical_to_oracle_keyed(name, tmp_path):
    """-ap 0 in the keyed schedule (layout in include/instruct_hip.h: a (cluster, locus) slot holds both subgenomes' Dirichlets,
    loci with four distinct alleles count as drawn): every sweep equal to the oracle's keyed chain"""
    from instruct_amd import synth
    extra = name in ALLO_EXTRA
    cfg = ALLO_EXTRA[name] if extra else ALLO[name]
    raw = synth.raw_alleles(cfg[0], cfg[1], cfg[2], 4, cfg[3], cfg[4], 20260401 + sorted(ALLO_EXTRA).index(name)) if extra else gu.make_golden.allo_data_for(name)
    txt = str(tmp_path / (name + ".txt"))
    synth.write_text_polyploid(txt, raw)
    N, L, K, A, miss, u, b, t, e, r, j, seeds = cfg
    out = str(tmp_path / (name + ".key"))
    assert subprocess.call([DUMP, txt, out] + [str(x) for x in (K, N, L, u, b, t, e, r, j) + tuple(seeds)] + ["1", "1", "1", "1"]) == 0
    want = [l for l in gu.parse(out) if l.startswith("it ") or l.startswith("chain zqinit")]
    got = hip_lines(name, cfg, raw, sched=1, allo=True)
    assert len(got) == len(want) == 1 + 6 * u
    for g, w in zip(got, want):
        assert _norm(_noseeds(g)) == _norm(_noseeds(w))


def _noseeds(line):
    return line.split(" seeds=")[0]


@pytest.mark.parametrize("sched", [0, 1])
@pytest.mark.parametrize("name", sorted(EXTRA))
def test_tetraploid_generated_cases_bit_identical_to_canonical_oracle(name, sched, tmp_path):
    """sched 1 = keyed schedule (every consumer seeks to its own stream position; the sequential seed triple is
    not defined there and is left out of the comparison)"""
    from instruct_amd import synth
    N, L, K, A, miss, u, b, t, e, r, j, seeds = EXTRA[name]
    raw = extra_data(name)
    txt, out = str(tmp_path / (name + ".txt")), str(tmp_path / (name + ".can"))
    synth.write_text_polyploid(txt, raw)
    args = [DUMP, txt, out] + [str(x) for x in (K, N, L, u, b, t, e, r, j) + tuple(seeds)] + ["1", "1"] + (["1"] if sched else [])
    assert subprocess.call(args) == 0
    want = [l for l in gu.parse(out) if l.startswith("it ") or l.startswith("chain zqinit")]
    got = hip_lines(name, EXTRA[name], raw, sched)
    assert len(got) == len(want) and len(got) == 1 + 6 * u
    for g, w in zip(got, want):
        if sched:
            g, w = _noseeds(g), _noseeds(w)
        assert _norm(g) == _norm(w)


@pytest.mark.parametrize("name", sorted(POLY))
def test_tetraploid_keyed_schedule_bit_identical_to_oracle_keyed(name, tmp_path):
    N, L, K, A, miss, u, b, t, e, r, j, seeds = POLY[name]
    out = str(tmp_path / (name + ".key"))
    args = [DUMP, os.path.join(gu.GOLDEN, name + ".txt"), out] + [str(x) for x in (K, N, L, u, b, t, e, r, j) + tuple(seeds)] + ["1", "1", "1"]
    assert subprocess.call(args) == 0
    want = [l for l in gu.parse(out) if l.startswith("it ") or l.startswith("chain zqinit")]
    got = hip_lines(name, sched=1)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert _norm(_noseeds(g)) == _norm(_noseeds(w))


@pytest.mark.parametrize("name", sorted(POLY))
def test_tetraploid_matches_reference_golden(name):
    want = [l for l in gu.parse(os.path.join(gu.GOLDEN, name + ".golden")) if l.startswith("it ") or l.startswith("chain zqinit")]
    got = hip_lines(name)
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


# ---------------------------------------------------------------------------------------------- BASELINE config 5
C5 = dict(N=10000, L=20000, K=10, A=4, miss=0.05)


def test_config5_shape_bit_identical_to_canonical_oracle(tmp_path):
    """Config 5's row shape (L=20000 -> 79 workgroups per individual, K=10 -> two count words per cluster group,
    4 alleles, 5 % missing) at the largest N the CPU oracle finishes in about a minute (N=320, 2 iterations):
    every dump line of the replay schedule identical to the canonical oracle."""
    from instruct_amd import synth
    cfg = (320, C5["L"], C5["K"], C5["A"], C5["miss"], 2, 1, 1, 1, 1, 1, (13, 4, 1972))
    raw = synth.raw_alleles(cfg[0], cfg[1], cfg[2], 4, cfg[3], cfg[4], 20260105)
    txt, out = str(tmp_path / "c5s.txt"), str(tmp_path / "c5s.can")
    synth.write_text_polyploid(txt, raw)
    N, L, K, A, miss, u, b, t, e, r, j, seeds = cfg
    args = [DUMP, txt, out] + [str(x) for x in (K, N, L, u, b, t, e, r, j) + tuple(seeds)] + ["1", "1"]
    assert subprocess.call(args) == 0
    want = [l for l in gu.parse(out) if l.startswith("it ") or l.startswith("chain zqinit")]
    got = hip_lines("c5s", cfg, raw, 0, fast_coder=True)
    assert len(got) == len(want) == 1 + 6 * u
    for g, w in zip(got, want):
        assert _norm(g) == _norm(w)


@pytest.fixture(scope="module")
def config5_data():
    """N=10000 L=20000 4 alleles 5 % missing: 1000 distinct individuals x 10 replicas (as bench.py's ploidy-4 leg;
    every replica is sampled independently by the chain)"""
    from instruct_amd import synth
    base = 1000
    raw = synth.raw_alleles(base, C5["L"], C5["K"], 4, C5["A"], C5["miss"], 20260105)
    obs, alleleid, allelenum = synth.code_tetraploid_fast(raw)
    rep = C5["N"] // base
    return np.tile(obs, (rep, 1, 1)), np.tile(alleleid, (rep, 1)), allelenum


def _check_config5_state(ch, obs, alleleid, K):
    """size-independent properties of the ploidy-4 sampler state (any N, L)"""
    N, L = alleleid.shape
    valid = alleleid > 0
    z = ch.z()
    assert (z[valid] >= 0).all() and (z[valid] < K).all() and (z[~valid] == -1).all()
    g = ch.geno()
    assert (g[~valid] == -1).all()
    # the imputed genotype is a sorted 4-copy multiset over exactly the observed allele set (initial_geno / update_geno,
    # poly_geno.c:316-369, 520-580): every observed allele present, nothing else present
    for i0 in range(0, N, 500):   # chunks of individuals bound the temporaries
        gb, ob, vb = g[i0:i0 + 500], obs[i0:i0 + 500], valid[i0:i0 + 500]
        gmask = np.bitwise_or.reduce(np.where(gb >= 0, 1 << np.maximum(gb, 0), 0), axis=2)
        omask = np.bitwise_or.reduce(np.where(ob >= 0, 1 << np.maximum(ob, 0), 0), axis=2)
        assert np.array_equal(gmask[vb], omask[vb]) and (gb[vb] >= 0).all()
    # allele counts == bincount of (z, locus, imputed allele) over the valid copies
    cnt = ch.count_alleles()
    A = ch.Amax
    idx = (z.astype(np.int64) * L + np.arange(L, dtype=np.int64)[None, :, None]) * A + g
    ref = np.bincount(idx[np.broadcast_to(valid[:, :, None], idx.shape)], minlength=K * L * A).reshape(K, L, A)
    assert np.array_equal(cnt, ref) and int(cnt.sum()) == 4 * int(valid.sum())
    del idx
    # qqnum rows = histogram of the individual's Z; qq rows are probability vectors
    qn = ch.qqnum()
    zz = np.where(valid[:, :, None], z, K).reshape(N, -1)
    assert np.array_equal(qn, np.stack([(zz == k).sum(1) for k in range(K)], 1).astype(float))
    qq = ch.qq()
    assert np.allclose(qq.sum(1), 1.0, atol=1e-12) and (qq > 0).all()
    assert np.isfinite(ch.indvlkh()).all() and ch.totallkh() < 0
    s = ch.self_rates()
    assert ((s >= 0) & (s <= 1)).all()


@pytest.mark.parametrize("sched", [0, 1])
def test_config5_full_size_properties(config5_data, sched):
    """BASELINE config 5 (N=10000, L=20000, K=10, ploidy 4, 5 % missing) on the device, both schedules, 2 iterations:
    state invariants that hold at any size, and a second run with the same seeds ends in the same state and at the
    same stream position (replay: the seed triple; keyed: order-independent sums -> same bits)."""
    from instruct_amd import capi
    obs, alleleid, allelenum = config5_data
    K = C5["K"]
    sig = []
    for rep in range(2):
        ch = capi.HipPolyChain(obs, alleleid, allelenum, K, back_refl=1, rng_sched=sched)
        ch.setseeds(13, 4, 1972)
        ch.chain_init(np.array([np.float32(ch.ran1()) for _ in range(K)], dtype=np.float32))
        ch.run(2)
        if rep == 0:
            _check_config5_state(ch, obs, alleleid, K)
        sig.append((orc.fnv_i32(ch.z()), orc.fnv_i32(ch.geno()), orc.fnv_f64(ch.qq()), ch.totallkh(), ch.seeds()))
        ch.close()
    assert sig[0] == sig[1]


@pytest.mark.parametrize("strip", ["0", "1"])
def test_tetraploid_interval_resolver_and_device_update_P_take_the_sweeps(strip, monkeypatch):
    """the generated case x_spec (pinned to the oracle line by line above): update_ZQ settled by the interval resolver (isg_spec_hip.inc),
    update_P_auto's Dirichlets resolved on the device (isg_walk_hip.inc) -- not by their fallbacks.  strip = 1: the expected counts from the
    strip-per-wave kernel (k4_zexpect) instead of one lane per individual (k4_zexpect_ind)"""
    from instruct_amd import capi, synth
    monkeypatch.setenv("INSTRUCT_ZEXPECT_STRIP", strip)
    N, L, K, A, miss = EXTRA["x_spec"][:5]
    obs, alleleid, allelenum = synth.code_tetraploid(extra_data("x_spec"))
    ch = capi.HipPolyChain(obs, alleleid, allelenum, K)
    ch.setseeds(28, 14, 2006)
    ch.chain_init(np.array([np.float32(ch.ran1()) for _ in range(K)], dtype=np.float32))
    for _ in range(4):
        ch.iteration()
    zs, pd = ch.zq_spec_stats(), ch.p_device_stats()
    ch.close()
    assert zs["settled"] >= 3 and zs["lost"] == 0, zs
    assert pd["device_sweeps"] == 4 and pd["host_sweeps"] == 0, pd
