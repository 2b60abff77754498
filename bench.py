#!/usr/bin/env python3
"""Headline benchmark: MCMC iterations/s of the InStruct hot path on MI355X.

A "step" is one MCMC iteration of one chain = the reference loop body mcmc.c:210-215
(update_P + update_S_POP + update_G + update_ZQ + update_alpha + cal_lkh), steady state, with the
packed genotypes and the sampler state already resident in HBM.  Workload: BASELINE.json config 3,
N=10000 individuals x L=5000 loci, K=5, diploid, mode 2 (-v 2 -e 1 -y 1), synthetic data.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1, one chain per GPU)

`value` is measured on the REPLAY schedule (stream positions identical to the reference: Z, allele
counts, generations and seeds are bit-identical to the reference run with the same seeds), over EXACTLY
K steps with the per-kernel profiling OFF; a second, profiled pass of the same chain gives `kernels_ms`
and the roofline entries (HIP events on the launch stream).  The same JSON line carries the KEYED
schedule (counter-based positions, all consumers concurrent) under "keyed" and BASELINE config 5
(ploidy 4) under "ploidy4".  Rank 0 at N=1 also times the reference's own CPU sweeps on the host
cores ("cpu_baseline").
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from instruct_amd import capi, multichain, synth  # noqa: E402

WORKLOADS = {
    "c3": dict(N=10000, L=5000, K=5, name="config3: N=10000 L=5000 K=5 diploid mode 2, 1 chain per GPU"),
    "c2": dict(N=2000, L=1000, K=5, name="config2: N=2000 L=1000 K=5 diploid mode 2, 1 chain per GPU"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILED_STEPS = 20    # the second pass: per-kernel times


def sync():
    import torch
    torch.cuda.synchronize()


def note(msg):
    """progress on stderr (stdout carries the ONE JSON line): which leg is running, should the process ever die inside one"""
    print("[bench %.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


def timed_steps(chain, steps, warmup, world):
    """(seconds for exactly `steps` iterations, profiling off; log-likelihood samples; per-kernel profile of a second pass)"""
    import torch.distributed as dist
    chain.run(warmup)
    chain.profile(False)
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    chain.run(steps)            # EXACTLY `steps` iterations: one C call, state stays on the device
    last = chain.totallkh()     # drains the stream (the last cal_lkh result comes back)
    sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = multichain.max_over_ranks(dt)
    # second pass, same chain, per-kernel HIP events on (not part of `value`)
    chain.profile_reset()
    chain.profile(True)
    chain.run(PROFILED_STEPS)
    chain.totallkh()
    chain.profile(False)
    prof = chain.profile_results()
    # log-likelihood samples for the Gelman-Rubin exchange: a few more (untimed) stored iterations
    lk = [last]
    for _ in range(5):
        chain.iteration()
        lk.append(chain.totallkh())
    return dt, lk, prof


# update_ZQ in the replay schedule is a PHASE of kernels.  Round 3 (isg_spec_hip.inc): expected cluster counts, accept-bit tables and walks
# that resolve every individual's start position, probes on the trajectory, then the sweep at the resolved positions (k_zq_at).  The block
# resolver of round 2 (k_tapef + k_zq_blocks / k_zq_block + k_zq_at) takes the sweeps the interval resolver hands on.
ZQ_PHASE = ("k_zexpect", "k_wk_centers_Z", "k_wk_table_Z", "k_wk_walk_Z", "k_zs_band", "k_zq_probe", "k_zs_offs", "k_zq_at", "k_tapef", "k_zq_blocks", "k_zq_block",
            "k_tape", "k_zq_pipe", "k_zq_spec", "k_zq_coop", "k_zq_chain", "k4_zq_coop", "k4_zq", "k_zq_keyed", "k4_zq_keyed")


def kernel_alg_bytes(name, N, L, P, nvalid_copies, extra):
    """ALGORITHMIC bytes of one launch of a kernel: what it must read and write, not what it happens to move."""
    cells = nvalid_copies if nvalid_copies else N * L * P
    table = {
        # sweeps over the genotype / Z bytes
        "k_zq_at": 2 * cells, "k_zq_keyed": 2 * cells, "k4_zq_keyed": 2 * cells, "k4_zq": 2 * cells, "k4_zq_coop": 2 * cells,
        "k_zq_blocks": 2 * cells, "k_zq_pipe": 2 * cells, "k_zq_chain": 2 * cells,
        "k_zexpect": cells,                                   # the genotype byte of every copy
        "k_count": 2 * cells, "k_loglik_pair": 2 * cells, "k_loglik_lkh": 2 * cells,
        "k4_geno": 3 * cells + cells, "k4_lkd": 3 * cells, "k4_sweep_counts": 2 * cells,
        # the accept-bit tables: one byte written per (group, window column); inputs are a few bytes per gamma
        "k_wk_table_Z": extra.get("table_bytes_Z", 0), "k_wk_table_P": extra.get("table_bytes_P", 0),
    }
    return table.get(name)


def roofline_of(prof, steps, N, L, P, nvalid_copies, extra, traffic, phase_bytes):
    """the dominant kernel of the profiled pass (largest total time) against the HBM roofline, and the update_ZQ phase as a sub-entry"""
    tot = {k: ms for k, (ms, n) in prof.items()}
    dom = max(tot, key=tot.get)
    ms, n = prof[dom]
    empty = 0

    def real_launches(k, cnt):
        # k_zq_at is also launched behind every blind probe round of the interval resolver and returns at once while the trajectory still
        # holds uncertain bytes (a few microseconds): ONE launch per iteration is the sweep, the average is taken over those
        return steps if (k == "k_zq_at" and cnt > steps) else cnt

    if real_launches(dom, n) != n:
        empty = n - steps
        n = steps
    alg = kernel_alg_bytes(dom, N, L, P, nvalid_copies, extra)
    avg_s = ms / n * 1e-3
    out = {"bound": "hbm", "kernel": dom, "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": round(ms / n, 4), "launches_per_step": round(n / steps, 2),
           "ms_per_step": round(ms / steps, 4), "alg_bytes_per_launch": alg,
           "traffic": traffic.get(dom), "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs)"}
    if alg:
        gbs = alg / avg_s / 1e9
        out.update(achieved=round(gbs, 3), frac=round(gbs / HBM_PEAK_GBS, 6))
    else:
        out.update(achieved=None, frac=None)
    if empty:
        out["early_exit_launches_per_step"] = round(empty / steps, 2)
    if dom.startswith("k_wk_table"):
        out["note"] = ("accept-bit tables of the walk engine: VALU work (one rgamma attempt decision per (gamma, window column)), hardly any HBM traffic -- "
                       "its algorithmic bytes are the table bytes it writes; the streaming sweep is `sweep_kernel`, the whole phase `phase`")
    ph = [k for k in ZQ_PHASE if k in prof]
    ph_ms = sum(prof[k][0] for k in ph) / steps
    out["phase"] = {"name": "update_ZQ: " + " + ".join(ph), "ms_per_step": round(ph_ms, 4), "alg_bytes": phase_bytes,
                    "achieved": round(phase_bytes / (ph_ms * 1e-3) / 1e9, 3), "frac": round(phase_bytes / (ph_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                    "traffic": traffic.get("update_ZQ_replay")}
    sk = next((k for k in ("k_zq_at", "k_zq_keyed", "k4_zq_keyed", "k4_zq_coop", "k4_zq") if k in prof), None)
    if sk:
        sms, sn = prof[sk]
        sn = real_launches(sk, sn)
        sb = kernel_alg_bytes(sk, N, L, P, nvalid_copies, extra)
        out["sweep_kernel"] = {"kernel": sk, "avg_launch_ms": round(sms / sn, 4), "alg_bytes_per_launch": sb,
                               "achieved": round(sb / (sms / sn * 1e-3) / 1e9, 3), "frac": round(sb / (sms / sn * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                               "traffic": traffic.get(sk)}
    return out


def load_traffic():
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        with open(p) as f:
            return json.load(f)
    return {}


def cpu_baseline(geno, K, seeds):
    """The reference's own sweeps on one host core (the reference is single threaded):
    oracle/_ref/ref_bench = reference mcmc.c compiled from /root/reference ("reference"); if that binary
    did not travel, the CPU restatement oracle/liborc.so in its reference configuration ("port")."""
    N, L, _ = geno.shape
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    iters = 3 if N * L >= 10_000_000 else 5
    sample = (f"chain init + {iters} full iterations at N={N} L={L} K={K} (same data, same seeds), every iteration timed by itself; "
              "value = 1 / median")
    if os.path.exists(exe):
        with tempfile.NamedTemporaryFile(suffix=".u8", delete=False) as f:
            np.where(geno < 0, 255, geno).astype(np.uint8).tofile(f)
            path = f.name
        try:
            p = subprocess.run([exe, path, str(N), str(L), str(K), str(iters)] + [str(s) for s in seeds], capture_output=True, timeout=1500)
            r = json.loads(p.stderr.decode().strip().splitlines()[-1])
        finally:
            os.unlink(path)
        per = sorted(r["per_iter_s"])
        med = per[len(per) // 2]
        return {"value": round(1.0 / med, 6), "unit": "iterations/s", "cores": 1, "kind": "reference", "sample": sample,
                "s_per_iter": med, "s_per_iter_min": per[0], "s_per_iter_median": med, "s_per_iter_all": r["per_iter_s"],
                "sweeps_s": {k: r[k] for k in ("update_P", "update_S_POP", "update_G", "update_ZQ", "update_alpha", "cal_lkh")}}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    an = np.full(L, int(geno.max()) + 1, dtype=np.int32)
    o = orc.OrcChain(geno, an, (geno < 0).any(-1).astype(np.int32), K)
    o.setseeds(*seeds)
    o.chain_init(np.array([o.ran1() for _ in range(K)], dtype=np.float32))
    t0 = time.perf_counter()
    o.iteration()
    dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 6), "unit": "iterations/s", "cores": 1, "kind": "port", "sample": sample, "s_per_iter": dt}


def cpu_baseline_concurrent(geno, K, seeds, nproc=8):
    """what a host gives the 8-chain configurations -- `nproc` single-chain reference processes side by side
    (the reference is single threaded: `-c 8` runs its chains back to back; eight processes are the best a user can do)."""
    N, L, _ = geno.shape
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    if not os.path.exists(exe):
        return None
    nproc = max(1, min(nproc, os.cpu_count() or 1))
    iters = 3
    with tempfile.NamedTemporaryFile(suffix=".u8", delete=False) as f:
        np.where(geno < 0, 255, geno).astype(np.uint8).tofile(f)
        path = f.name
    try:
        t0 = time.perf_counter()
        procs = [subprocess.Popen([exe, path, str(N), str(L), str(K), str(iters)] + [str(s) for s in multichain.rank_seeds(seeds, r)],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.PIPE) for r in range(nproc)]
        outs = [p.communicate(timeout=1500)[1] for p in procs]
        wall = time.perf_counter() - t0
    finally:
        os.unlink(path)
    meds = []
    for o in outs:
        per = sorted(json.loads(o.decode().strip().splitlines()[-1])["per_iter_s"])
        meds.append(per[len(per) // 2])
    return {"value": round(sum(1.0 / m for m in meds), 6), "unit": "chain-iterations/s", "cores": nproc, "kind": "reference",
            "sample": f"{nproc} reference processes at once (seeds of ranks 0..{nproc - 1}), chain init + {iters} iterations each at N={N} L={L} K={K}; "
                      "value = sum over processes of 1 / median iteration time",
            "s_per_iter_median_each": [round(m, 4) for m in meds], "wall_s": round(wall, 1)}


def chain_worker(workload, rank, steps, device, rendezvous):
    """a worker PROCESS of the several-chains-on-one-GPU leg: its own context on `device`, waits for the go file, runs `steps` iterations"""
    w = WORKLOADS[workload]
    geno, an, mi = synth.make_diploid(w["N"], w["L"], w["K"])
    h = capi.HipChain(geno, an, mi, w["K"], rng_sched=capi.SCHED_REPLAY, device=device)
    h.setseeds(*multichain.rank_seeds((13, 4, 1972), rank))
    h.chain_init(np.array([h.ran1() for _ in range(w["K"])], dtype=np.float32))
    h.run(2)
    h.totallkh()
    open(os.path.join(rendezvous, "ready.%d" % rank), "w").close()
    go = os.path.join(rendezvous, "go")
    t_wait = time.time()
    while not os.path.exists(go):
        if time.time() - t_wait > 600:
            raise SystemExit("worker %d: no go file" % rank)
        time.sleep(0.002)
    t0 = time.perf_counter()
    h.run(steps)
    h.totallkh()
    dt = time.perf_counter() - t0
    print(json.dumps({"rank": rank, "s": dt, "start": t0, "end": t0 + dt}), flush=True)
    h.close()


def concurrent_chains_leg(geno, an, mi, K, device, steps, workload):
    """Independent replay chains sharing ONE GPU, seeds per chain as for the multi-GPU runs: (a) one host thread + one HIP stream per chain in
    this process, (b) one PROCESS per chain (what the multi-GPU launcher starts when chains outnumber GPUs; each with its own HIP context and
    its own host threads) -- the only evidence about host-side contention obtainable on a one-GPU box."""
    import threading
    res = {}
    chains = []
    for r in range(8):
        h = capi.HipChain(geno, an, mi, K, rng_sched=capi.SCHED_REPLAY, device=device)
        h.setseeds(*multichain.rank_seeds((13, 4, 1972), r))
        h.chain_init(np.array([h.ran1() for _ in range(K)], dtype=np.float32))
        h.run(1)
        chains.append(h)

    def work(h):
        h.run(steps)
        h.totallkh()
    for n in (2, 8):
        th = [threading.Thread(target=work, args=(h,)) for h in chains[:n]]
        sync()
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        sync()
        dt = time.perf_counter() - t0
        res[str(n)] = {"chain_iterations_per_s": round(n * steps / dt, 3), "ms_per_step_per_chain": round(dt / steps * 1e3, 3)}
    for h in chains:
        h.close()
    res["note"] = "aggregate over n chains on one GPU (threads of this process), replay schedule; not part of `value` (1 chain per GPU)"
    # (b) processes: 4 workers (the test boxes allow at most 6 processes on a GPU at once, this one included)
    nproc = 4
    note("several chains on one GPU: %d processes" % nproc)
    with tempfile.TemporaryDirectory() as rv:
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--chain-worker", str(r), "--steps", str(steps), "--workload", workload,
                                   "--device", str(device), "--rendezvous", rv], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for r in range(nproc)]
        t_wait = time.time()
        while sum(os.path.exists(os.path.join(rv, "ready.%d" % r)) for r in range(nproc)) < nproc and time.time() - t_wait < 600:
            if any(p.poll() is not None for p in procs):
                break
            time.sleep(0.01)
        open(os.path.join(rv, "go"), "w").close()
        outs = []
        for p in procs:
            o, _ = p.communicate(timeout=900)
            try:
                outs.append(json.loads(o.decode().strip().splitlines()[-1]))
            except Exception:
                outs = None
                break
    if outs:
        span = max(o["s"] for o in outs)
        res["processes"] = {"n": nproc, "chain_iterations_per_s": round(nproc * steps / span, 3), "ms_per_step_per_chain": round(span / steps * 1e3, 3),
                            "note": "one process per chain on one GPU (own HIP context, own host threads), aggregate over the slowest worker's time"}
    return res


def tetra_leg(device, steps, warmup, with_cpu):
    """BASELINE.json config 5 on one GPU: N=10000 L=20000 K=10 ploidy 4 (autotetraploid), 5 % missing, replay schedule; 10000 DISTINCT
    synthetic individuals (instruct_amd/host/synth_fast.c: the numpy generator's arrays, built in seconds)."""
    N, L, K, A = 10000, 20000, 10, 4
    obs, alleleid, allelenum = synth.make_tetraploid_fast(N, L, K, A, 0.05, 20260105)
    nvalid = int((alleleid > 0).sum())
    traffic = load_traffic()
    # the ploidy-4 kernels are timed under the diploid phase names; their counter traffic is filed under their own kernel names (profiles/traffic.json)
    traffic4 = dict(traffic)
    for name, own in (("k_zq_at", "k4_zq_keyed"), ("k_zexpect", "k4_zexpect"), ("k_zq_probe", "k4_zq_probe")):
        traffic4[name] = traffic.get(own)
    traffic4["update_ZQ_replay"] = (sum(traffic[k] for k in ("k4_zexpect", "k4_zexpect_fin", "k4_zq_keyed", "k4_zq_probe", "k_wk_table_Z")) if all(
        k in traffic for k in ("k4_zexpect", "k4_zexpect_fin", "k4_zq_keyed", "k4_zq_probe", "k_wk_table_Z")) else None)

    def one(sched, nsteps):
        ch = capi.HipPolyChain(obs, alleleid, allelenum, K, back_refl=1, rng_sched=sched, device=device)
        ch.setseeds(13, 4, 1972)
        ch.chain_init(np.array([np.float32(ch.ran1()) for _ in range(K)], dtype=np.float32))
        ch.run(warmup)
        ch.profile(False)
        sync()
        t0 = time.perf_counter()
        ch.run(nsteps)
        last = ch.totallkh()
        sync()
        dt = time.perf_counter() - t0
        psteps = max(2, nsteps // 2)
        ch.profile_reset()
        ch.profile(True)
        ch.run(psteps)
        ch.totallkh()
        ch.profile(False)
        prof = ch.profile_results()
        extra = {"table_bytes_Z": ch.zq_spec_stats().get("table_bytes", 0), "table_bytes_P": ch.p_device_stats()["table_bytes"]}
        stats = {"interval_resolver": ch.zq_spec_stats(), "update_P_device": ch.p_device_stats(), "fallback_sweeps": ch.zq_fallbacks()} if sched == capi.SCHED_REPLAY else {}
        ch.close()
        out = {"value": round(nsteps / dt, 4), "unit": "iterations/s", "ms_per_step": round(dt / nsteps * 1e3, 3),
               "roofline": roofline_of(prof, psteps, N, L, 4, 4 * nvalid, extra, traffic4, 2 * 4 * nvalid),
               "kernels_ms": {k: round(ms / n, 4) for k, (ms, n) in sorted(prof.items())},
               "kernels_ms_per_step": {k: round(ms / psteps, 4) for k, (ms, n) in sorted(prof.items())},
               "iteration_frac_of_hbm": round(5 * N * L * 4 / (dt / nsteps) / 1e9 / HBM_PEAK_GBS, 6), "last_totallkh": last}
        out.update(stats)
        return out
    res = {"workload": "config5: N=10000 L=20000 K=10 ploidy 4 (-p 4 -ap 1), 5% missing, 10000 distinct individuals, 1 chain; value = replay schedule"}
    res.update(one(capi.SCHED_REPLAY, steps))
    res["keyed"] = one(capi.SCHED_KEYED, steps)
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bench_poly")
    if with_cpu and os.path.exists(exe):
        # the reference's own ploidy-4 sweeps (poly_geno.c:98-116, bare loop: nothing but the sweeps between the clock reads)
        # on one host core at 1/100 of the (individual, locus) cells: 3 iterations, each timed by itself
        n, l, iters = 500, 4000, 3
        o2, _, _ = synth.code_tetraploid_fast(synth.raw_alleles(n, l, K, 4, A, 0.05, 20260105))
        with tempfile.NamedTemporaryFile(suffix=".u8", delete=False) as f:
            np.where(o2 < 0, 255, o2).astype(np.uint8).tofile(f)
            path = f.name
        try:
            p = subprocess.run([exe, path, str(n), str(l), str(K), str(iters), "1", "13", "4", "1972"], capture_output=True, timeout=1500)
            r = json.loads(p.stderr.decode().strip().splitlines()[-1])
        finally:
            os.unlink(path)
        s_small = r["s_per_iter"]
        scale = (N * L) / (n * l)
        res["cpu_baseline"] = {"value": round(1.0 / (s_small * scale), 8), "unit": "iterations/s", "cores": 1, "kind": "reference",
                               "sample": f"reference poly_geno.c sweeps (bare loop) at N={n} L={l} K={K}, {iters} iterations, {s_small:.3f} s/iteration, "
                                         f"scaled by N*L = x{scale:.0f} (every sweep is linear in N*L)",
                               "s_per_iter_extrapolated": round(s_small * scale, 1),
                               "sweeps_s_small": {k: r[k] for k in ("update_P", "update_S_POP", "update_ZQ", "update_geno", "cal_lkd")}}
        res["speedup_vs_cpu"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
        res["keyed"]["speedup_vs_cpu"] = round(res["keyed"]["value"] / res["cpu_baseline"]["value"], 1)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-keyed", action="store_true")
    ap.add_argument("--no-tetra", action="store_true", help="skip the ploidy 4 (config 5) leg")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the several-chains-on-one-GPU leg (profiling: keeps per-kernel averages single-chain)")
    ap.add_argument("--chain-worker", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--rendezvous", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.chain_worker is not None:
        return chain_worker(args.workload, args.chain_worker, args.steps, args.device, args.rendezvous)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("ISG_BENCH_BACKEND", "nccl")  # "gloo": rehearse N ranks on fewer GPUs (tests only)
    if backend != "nccl":
        local = local % ndev
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    w = WORKLOADS[args.workload]
    K = w["K"]
    geno, an, mi = synth.make_diploid(w["N"], w["L"], K)
    N, L, P = geno.shape
    seeds = multichain.rank_seeds((13, 4, 1972), rank)
    traffic = load_traffic()
    out = {}
    scheds = [("replay", capi.SCHED_REPLAY)] + ([] if args.no_keyed else [("keyed", capi.SCHED_KEYED)])
    for tag, sched in scheds:
        if rank == 0:
            note(tag + " schedule")
        ch = capi.HipChain(geno, an, mi, K, mode=2, type_freq=1, back_refl=1, rng_sched=sched, device=local)
        ch.setseeds(*seeds)
        ch.chain_init(np.array([ch.ran1() for _ in range(K)], dtype=np.float32))
        dt, lk, prof = timed_steps(ch, args.steps, args.warmup, world)
        extra = {"table_bytes_Z": ch.zq_spec_stats().get("table_bytes", 0), "table_bytes_P": ch.p_device_stats()["table_bytes"]}
        rl = roofline_of(prof, PROFILED_STEPS, N, L, P, 0, extra, traffic, 2 * N * L * P)
        if sched == capi.SCHED_REPLAY:
            rl["interval_resolver"] = ch.zq_spec_stats()
            rl["update_P_device"] = ch.p_device_stats()
            rl["block_resolver"] = ch.zq_resolve_stats()
            rl["fallback_sweeps"] = ch.zq_fallbacks()
        ckrep = min(len(lk), 20)
        gr = None
        if world > 1:
            gr = multichain.gelman_rubin_all_ranks(np.array(lk[-ckrep:]))  # RCCL all-gather over xGMI
        out[tag] = dict(value=world * args.steps / dt, ms_per_step=dt / args.steps * 1e3, roofline=rl, gelman_rubin=gr,
                        kernels_ms={k: round(ms / n, 4) for k, (ms, n) in sorted(prof.items())},
                        kernels_ms_per_step={k: round(ms / PROFILED_STEPS, 4) for k, (ms, n) in sorted(prof.items())},
                        # whole iteration against the fused-design contract figure 3*N*L*P bytes
                        iteration_frac_of_hbm=round(3 * N * L * P / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 6),
                        last_totallkh=lk[-1])
        ch.close()

    if rank == 0:
        cpu = cpu8 = None
        if world == 1 and not args.no_cpu:
            note("cpu baseline: one reference process")
            cpu = cpu_baseline(geno, K, seeds)
            note("cpu baseline: eight reference processes")
            cpu8 = cpu_baseline_concurrent(geno, K, seeds)
        note("copy bandwidth")
        head = out["replay"]
        copy_gbs = round(capi.copy_bandwidth(local), 1)  # the fraction against a measured device-to-device copy as well (16-byte accesses)
        for o in out.values():
            o["roofline"]["copy_peak_measured"] = copy_gbs
            if o["roofline"].get("achieved"):
                o["roofline"]["frac_of_measured_copy"] = round(o["roofline"]["achieved"] / copy_gbs, 6)
        line = {
            "metric": "MCMC iterations/sec (update_P+update_ZQ+update_SG) at NxLxK", "value": round(head["value"], 4),
            "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(head["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": w["name"], "N": N, "L": L, "K": K, "ploidy": P, "mode": 2, "rng_schedule": "replay",
                       "chains": world, "parallelism": f"{world} independent chain(s), one per GPU"},
            "roofline": head["roofline"], "cpu_baseline": cpu, "cpu_baseline_8_processes": cpu8,
            "speedup_vs_cpu": (round(head["value"] / world / cpu["value"], 2) if cpu else None),
            "kernels_ms": head["kernels_ms"], "kernels_ms_per_step": head["kernels_ms_per_step"], "iteration_frac_of_hbm": head["iteration_frac_of_hbm"],
            "gelman_rubin": head["gelman_rubin"],
            "timing": f"value: exactly {args.steps} iterations after {args.warmup} warm-up, per-kernel events off; kernels_ms / roofline: a second pass of {PROFILED_STEPS} iterations with events on",
        }
        if "keyed" in out:
            k = out["keyed"]
            line["keyed"] = {"value": round(k["value"], 4), "unit": "iterations/s", "ms_per_step": round(k["ms_per_step"], 4),
                             "roofline": k["roofline"], "kernels_ms": k["kernels_ms"], "iteration_frac_of_hbm": k["iteration_frac_of_hbm"],
                             "speedup_vs_cpu": (round(k["value"] / world / cpu["value"], 2) if cpu else None),
                             "gelman_rubin": k["gelman_rubin"],
                             "note": "counter-based stream positions: bit-identical to the oracle's keyed schedule, statistically equivalent to the reference"}
        if world == 1 and not args.no_tetra:
            if not args.no_concurrent:
                note("several chains on one GPU: threads, then processes")
                line["concurrent_chains"] = concurrent_chains_leg(geno, an, mi, K, local, max(10, args.steps // 4), args.workload)
                if cpu8 and line["concurrent_chains"].get("processes"):  # several chains on ONE GPU against eight host cores
                    line["concurrent_chains"]["processes"]["vs_cpu_8_processes"] = round(line["concurrent_chains"]["processes"]["chain_iterations_per_s"] / cpu8["value"], 1)
            note("ploidy 4 (config 5)")
            line["ploidy4"] = tetra_leg(local, max(4, args.steps // 10), 2, not args.no_cpu)
        note("done")
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
