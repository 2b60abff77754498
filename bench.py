#!/usr/bin/env python3
"""Headline benchmark: MCMC iterations/s of the InStruct hot path on MI355X.

A "step" is one MCMC iteration of one chain = the reference loop body mcmc.c:210-215
(update_P + update_S_POP + update_G + update_ZQ + update_alpha + cal_lkh), steady state, with the
packed genotypes and the sampler state already resident in HBM.  Workload: BASELINE.json config 3,
N=10000 individuals x L=5000 loci, K=5, diploid, mode 2 (-v 2 -e 1 -y 1), synthetic data.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1, one chain per GPU)

`value` is measured on the REPLAY schedule (stream positions identical to the reference: Z, allele
counts, generations and seeds are bit-identical to the reference run with the same seeds).  The same
JSON line carries the KEYED schedule (counter-based positions, all consumers concurrent) under "keyed".
Rank 0 at N=1 also times the reference's own CPU sweeps on the host cores ("cpu_baseline").
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


def sync():
    import torch
    torch.cuda.synchronize()


def timed_steps(chain, steps, warmup, world):
    import torch.distributed as dist
    chain.run(warmup)
    lk = []
    chain.profile_reset()
    chain.profile(True)
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
    chain.profile(False)
    prof = chain.profile_results()
    if world > 1:
        dt = multichain.max_over_ranks(dt)
    # log-likelihood samples for the Gelman-Rubin exchange: a few more (untimed) stored iterations
    lk = [last]
    for _ in range(5):
        chain.iteration()
        lk.append(chain.totallkh())
    return dt, lk, prof


# the replay schedule's update_ZQ is a PHASE of kernels: the uniforms as floats, the resolution of the start positions block by block
# (k_zq_blocks: one launch, all blocks; k_zq_block: one launch per block when the workgroups cannot all be resident), then the
# sweep at the resolved positions.  Its roofline entry is the phase: 2 N L P bytes / phase time.
ZQ_RESOLVE = ("k_tapef", "k_zq_blocks", "k_zq_at")
ZQ_RESOLVE_PER_BLOCK = ("k_tapef", "k_zq_block", "k_zq_at")
# round 3: the start positions from intervals of the Dirichlets' shapes (isg_spec_hip.inc): expected counts, accept-bit tables, walks,
# probes on the trajectory, then the same sweep kernel
ZQ_SPEC = ("k_zexpect", "k_wk_centers_Z", "k_wk_table_Z", "k_wk_walk_Z", "k_zs_band", "k_zq_probe", "k_zs_offs", "k_zq_at")


def roofline(prof, kernel, bytes_per_launch, traffic):
    if isinstance(kernel, tuple):
        kernel = tuple(k for k in kernel if k in prof)
        ms = sum(prof[k][0] for k in kernel)
        n = prof[kernel[-1]][1]
        name = "update_ZQ phase: " + " + ".join(kernel)
    else:
        ms, n = prof[kernel]
        name = kernel
    avg_s = ms / n * 1e-3
    gbs = bytes_per_launch / avg_s / 1e9
    return {"bound": "hbm", "kernel": name, "achieved": round(gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 6), "avg_launch_ms": round(ms / n, 4), "alg_bytes_per_launch": bytes_per_launch,
            "traffic": traffic, "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs)"}


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


def concurrent_chains_leg(geno, an, mi, K, device, steps):
    """Independent replay chains sharing ONE GPU (one host thread + one HIP stream per chain, seeds per chain as for
    the multi-GPU runs): the replay update_ZQ kernel is latency bound on 20 of the 256 CUs, so chains overlap."""
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
    res["note"] = "aggregate over n chains on one GPU, replay schedule; not part of `value` (1 chain per GPU)"
    return res


def tetra_leg(device, steps, warmup, with_cpu):
    """BASELINE.json config 5 on one GPU: N=10000 L=20000 K=10 ploidy 4 (autotetraploid), 5 % missing, replay
    schedule.  The synthetic population is 1000 distinct individuals x 10 replicas (the generator's numpy
    coder takes minutes at 8e8 allele copies); every replica is sampled independently by the chain."""
    N, L, K, A, base = 10000, 20000, 10, 4, 1000
    raw = synth.raw_alleles(base, L, K, 4, A, 0.05, 20260105)
    obs, alleleid, allelenum = synth.code_tetraploid_fast(raw)
    obs, alleleid = np.tile(obs, (N // base, 1, 1)), np.tile(alleleid, (N // base, 1))
    nvalid = int((alleleid > 0).sum())
    traffic = load_traffic()

    def one(sched, nsteps):
        ch = capi.HipPolyChain(obs, alleleid, allelenum, K, back_refl=1, rng_sched=sched, device=device)
        ch.setseeds(13, 4, 1972)
        ch.chain_init(np.array([np.float32(ch.ran1()) for _ in range(K)], dtype=np.float32))
        ch.run(warmup)
        ch.profile_reset()
        ch.profile(True)
        sync()
        t0 = time.perf_counter()
        ch.run(nsteps)
        last = ch.totallkh()
        sync()
        dt = time.perf_counter() - t0
        ch.profile(False)
        prof = ch.profile_results()
        ch.close()
        zq = next(k for k in ("k_zq_at", "k4_zq_coop", "k4_zq_keyed", "k4_zq") if k in prof)  # (k_zq_at: the sweep at resolved positions, k4_zq's code)
        # update_ZQ launch, algorithmic bytes: genotype + Z byte per allele copy, in both schedules (the replay schedule's
        # uniform tape is traffic, not algorithm)
        alg = 2 * 4 * nvalid
        return {"value": round(nsteps / dt, 4), "unit": "iterations/s", "ms_per_step": round(dt / nsteps * 1e3, 3),
                "roofline": roofline(prof, zq, alg, traffic.get(zq)),
                "kernels_ms": {k: round(ms / n, 4) for k, (ms, n) in sorted(prof.items())},
                "iteration_frac_of_hbm": round(5 * N * L * 4 / (dt / nsteps) / 1e9 / HBM_PEAK_GBS, 6), "last_totallkh": last}
    res = {"workload": "config5: N=10000 L=20000 K=10 ploidy 4 (-p 4 -ap 1), 5% missing, 1 chain; value = replay schedule"}
    res.update(one(capi.SCHED_REPLAY, steps))
    res["keyed"] = one(capi.SCHED_KEYED, 4 * steps)
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


def measured_copy_gbs(dev):
    """HBM bandwidth of a plain device-to-device copy on this GPU (1 GiB, read + written bytes), torch's stream"""
    import torch
    n = 1 << 30
    x = torch.empty(n, dtype=torch.uint8, device=f"cuda:{dev}")
    y = torch.empty_like(x)
    x.zero_()
    for _ in range(2):
        y.copy_(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 8
    for _ in range(reps):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    return round(2.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-keyed", action="store_true")
    ap.add_argument("--no-tetra", action="store_true", help="skip the ploidy 4 (config 5) leg")
    ap.add_argument("--no-concurrent", action="store_true", help="skip the several-chains-on-one-GPU leg (profiling: keeps per-kernel averages single-chain)")
    args = ap.parse_args()

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
        ch = capi.HipChain(geno, an, mi, K, mode=2, type_freq=1, back_refl=1, rng_sched=sched, device=local)
        ch.setseeds(*seeds)
        ch.chain_init(np.array([ch.ran1() for _ in range(K)], dtype=np.float32))
        dt, lk, prof = timed_steps(ch, args.steps, args.warmup, world)
        if sched != capi.SCHED_REPLAY:
            zq = "k_zq_keyed"
        elif "k_zexpect" in prof:
            zq = ZQ_SPEC + tuple(k for k in ("k_tapef", "k_zq_blocks", "k_zq_block") if k in prof)  # (+ whatever sweeps fell through to the block resolver)
        elif "k_zq_blocks" in prof:
            zq = ZQ_RESOLVE
        elif "k_zq_block" in prof:
            zq = ZQ_RESOLVE_PER_BLOCK
        else:
            zq = next(k for k in ("k_zq_pipe", "k_zq_spec", "k_zq_coop", "k_zq_chain") if k in prof)
        # update_ZQ: reads the genotype byte and writes the Z byte of every allele copy
        rl = roofline(prof, zq, 2 * N * L * P, traffic.get("update_ZQ_replay" if isinstance(zq, tuple) else zq))
        if isinstance(zq, tuple):
            rl["resolve"] = ch.zq_resolve_stats()
            rl["interval_resolver"] = ch.zq_spec_stats()
            rl["update_P_device"] = ch.p_device_stats()
            rl["fallback_sweeps"] = ch.zq_fallbacks()
        ckrep = min(len(lk), 20)
        gr = None
        if world > 1:
            gr = multichain.gelman_rubin_all_ranks(np.array(lk[-ckrep:]))  # RCCL all-gather over xGMI
        out[tag] = dict(value=world * args.steps / dt, ms_per_step=dt / args.steps * 1e3, roofline=rl, gelman_rubin=gr,
                        kernels_ms={k: round(ms / n, 4) for k, (ms, n) in sorted(prof.items())},
                        # whole iteration against the fused-design contract figure 3*N*L*P bytes
                        iteration_frac_of_hbm=round(3 * N * L * P / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 6),
                        last_totallkh=lk[-1])
        ch.close()

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu:
            cpu = cpu_baseline(geno, K, seeds)
        head = out["replay"]
        copy_gbs = measured_copy_gbs(local)  # the fraction against a measured device-to-device copy as well
        for o in out.values():
            o["roofline"]["copy_peak_measured"] = copy_gbs
            o["roofline"]["frac_of_measured_copy"] = round(o["roofline"]["achieved"] / copy_gbs, 6)
        line = {
            "metric": "MCMC iterations/sec (update_P+update_ZQ+update_SG) at NxLxK", "value": round(head["value"], 4),
            "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(head["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": w["name"], "N": N, "L": L, "K": K, "ploidy": P, "mode": 2, "rng_schedule": "replay",
                       "chains": world, "parallelism": f"{world} independent chain(s), one per GPU"},
            "roofline": head["roofline"], "cpu_baseline": cpu,
            "speedup_vs_cpu": (round(head["value"] / world / cpu["value"], 2) if cpu else None),
            "kernels_ms": head["kernels_ms"], "iteration_frac_of_hbm": head["iteration_frac_of_hbm"],
            "gelman_rubin": head["gelman_rubin"],
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
                line["concurrent_chains"] = concurrent_chains_leg(geno, an, mi, K, local, max(4, args.steps // 2))
            line["ploidy4"] = tetra_leg(local, max(2, args.steps // 4), 1, not args.no_cpu)
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
