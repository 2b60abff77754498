#!/usr/bin/env python3
"""Research prototype: simulate the speculate / verify rounds of a replay update_ZQ resolver on the CPU.
Round 0 walks the start-position recurrence with APPROXIMATE Dirichlet shapes (previous iteration's counts), then
rounds of exact unit evaluations (UW candidates around the current trajectory) until the trajectory is covered."""
import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from spec_proto import wh_tape, dirich_consume, orc, synth

def main():
    N, L, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    iters = [int(x) for x in sys.argv[4].split(",")]
    UW = int(sys.argv[5]) if len(sys.argv) > 5 else 8
    LO = (UW - 1) // 2
    geno, an, mi = synth.make_diploid(N, L, K)
    N, L, P = geno.shape
    o = orc.OrcChain(geno, an, mi, K)
    o.setseeds(13, 4, 1972)
    o.chain_init(np.array([o.ran1() for _ in range(K)], dtype=np.float32))
    valid = o.valid().astype(bool); nval = valid.sum(1)
    prev_qqnum = o.qqnum().copy()
    for it in range(max(iters) + 1):
        o.update_P(); o.update_S_POP(); o.update_G()
        if it in iters:
            qq0 = o.qq().copy(); freq = o.freq().copy(); alpha = o.alpha(); seeds = o.seeds(); c0 = o.rng_count()
        o.update_ZQ(0)
        if it in iters:
            used = int(o.rng_count() - c0)
            U = wh_tape(seeds, used + 40000)
            B = np.concatenate([[0], np.cumsum(2 * nval + 2 * K)])
            thr = []
            for i in range(N):
                g = geno[i][valid[i]]; jj = np.nonzero(valid[i])[0]
                w = qq0[i][None, None, :] * freq[:, jj[:, None], g].transpose(1, 2, 0)
                cum = np.cumsum(w, -1); thr.append((cum / cum[..., -1:])[..., :-1].reshape(-1, K - 1))
            nexact = [0]
            def counts(i, e):
                nd = 2 * int(nval[i]); p = int(B[i]) + 2 * e
                z = (U[p:p + nd][:, None] > thr[i]).sum(1)
                return np.bincount(z, minlength=K).astype(float)
            def c_exact(i, e):
                nexact[0] += 1
                nd = 2 * int(nval[i]); p = int(B[i]) + 2 * e
                return (dirich_consume(U, p + nd, counts(i, e) + alpha) - 2 * K) // 2
            def c_approx(i, e, shapes):
                nd = 2 * int(nval[i]); p = int(B[i]) + 2 * e
                return (dirich_consume(U, p + nd, shapes + alpha) - 2 * K) // 2
            # true trajectory
            et = [0]
            for i in range(N): et.append(et[-1] + c_exact(i, et[-1]))
            assert B[N] + 2 * et[N] == used
            nexact[0] = 0
            shapes = prev_qqnum.copy()
            slot_lo = np.full(N, -10**9); slot_c = np.zeros((N, UW), dtype=int)
            rounds = []
            traj = None
            for rnd in range(40):
                e = 0; tr = []; unc = []
                for i in range(N):
                    tr.append(e)
                    if slot_lo[i] <= e < slot_lo[i] + UW: e += slot_c[i][e - slot_lo[i]]
                    else:
                        unc.append(i); e += c_approx(i, e, shapes[i])
                ncorrect = next((k for k in range(N) if tr[k] != et[k]), N)
                rounds.append((len(unc), ncorrect, int(np.abs(np.array(tr) - np.array(et[:N])).max())))
                if not unc: break
                for i in unc:
                    lo = max(0, tr[i] - LO)
                    slot_lo[i] = lo
                    for d in range(UW): slot_c[i][d] = c_exact(i, lo + d)
                    shapes[i] = counts(i, tr[i])
            assert tr == et[:N]
            print(f"iter {it}: alpha={alpha:.3f} qqmax={np.sort(qq0,1)[:,-1].mean():.3f} unit width {UW}: rounds (uncovered, correct prefix, max |dev|) = {rounds}"
                  f"  exact candidate evals/N = {nexact[0]/N:.2f}", flush=True)
        prev_qqnum = o.qqnum().copy()
        o.update_alpha(); o.cal_lkh()
if __name__ == "__main__":
    main()
