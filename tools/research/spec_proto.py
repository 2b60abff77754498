#!/usr/bin/env python3
"""Research prototype (CPU, numpy + the oracle): how well can the consumption c_i(e) of individual i's
Dirichlet in replay update_ZQ be predicted WITHOUT the exact cluster counts at start position e?
(mcmc.c:1122-1203, random.c:167-250).  Not product code."""
import os, sys, math
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
from instruct_amd import synth

E = 2.71828182

def wh_tape(seeds, n):
    """n uniforms after the state `seeds` (random.c:19-47)"""
    out = np.zeros(n)
    for s, a, m in zip(seeds, (171, 172, 170), (30269, 30307, 30323)):
        # s_k = s * a^k mod m, k = 1..n
        blk = 4096
        pw = np.empty(blk, dtype=np.int64); v = 1
        for k in range(blk):
            v = v * a % m; pw[k] = v
        ab = pw[-1]
        st = np.empty(n, dtype=np.int64)
        base = s % m
        for b0 in range(0, n, blk):
            l = min(blk, n - b0)
            st[b0:b0 + l] = base * pw[:l] % m
            base = base * ab % m
        out += st / float(m)
    return np.fmod(out, 1.0)

def gamma_try(U, pos, a):
    """one rgamma attempt at tape position pos; returns (accepted, newpos)"""
    if a < 1:
        u0, u1 = U[pos], U[pos + 1]; pos += 2
        if u0 > E / (a + E):
            r = -math.log((a + E) * (1 - u0) / (a * E))
            return (not (u1 > r ** (a - 1))), pos
        x = (a + E) * u0 / E
        r = x ** (1 / a)
        return (not (u1 > math.exp(-r))), pos
    if a == 1:
        return True, pos + 1
    c1 = a - 1; c2 = (a - 1 / (6 * a)) / c1; c3 = 2 / c1; c4 = c3 + 2; c5 = 1 / math.sqrt(a)
    while True:
        u1, u2 = U[pos], U[pos + 1]; pos += 2
        if a > 2.5: u1 = u2 + c5 * (1 - 1.86 * u1)
        if not (u1 >= 1 or u1 <= 0): break
    w = c2 * u2 / u1
    if c3 * u1 + w + 1 / w > c4:
        if c3 * math.log(u1) - math.log(w) + w >= 1: return False, pos
    return True, pos

def dirich_consume(U, pos, shapes):
    p0 = pos
    for a in shapes:
        while True:
            ok, pos = gamma_try(U, pos, a)
            if ok: break
    return pos - p0

def main():
    N, L, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    iters = [int(x) for x in sys.argv[4].split(",")]
    W = 6
    geno, an, mi = synth.make_diploid(N, L, K)
    N, L, P = geno.shape
    o = orc.OrcChain(geno, an, mi, K)
    o.setseeds(13, 4, 1972)
    o.chain_init(np.array([o.ran1() for _ in range(K)], dtype=np.float32))
    valid = o.valid().astype(bool)
    nval = valid.sum(1)
    prev_qqnum = o.qqnum().copy()
    for it in range(max(iters) + 1):
        o.update_P(); o.update_S_POP(); o.update_G()
        if it in iters:
            qq0 = o.qq().copy(); freq = o.freq().copy(); alpha = o.alpha(); seeds = o.seeds(); c0 = o.rng_count()
        o.update_ZQ(0)
        if it in iters:
            used = o.rng_count() - c0
            U = wh_tape(seeds, int(used) + 4000)
            qn = o.qqnum().copy()
            # true trajectory
            pos = 0; stats = dict(n=0, mm_prev=0, mm_nb=0, mm_exp=0, mm_sparse=0, cs=[]) ; starts = []
            rng = np.random.default_rng(1)
            sample = set(rng.choice(N, size=min(N, 250), replace=False).tolist())
            for i in range(N):
                starts.append(pos)
                nd = 2 * int(nval[i])
                shapes_true = qn[i] + alpha
                cons = dirich_consume(U, pos + nd, shapes_true)
                if i in sample and pos >= 2 * W:
                    g = geno[i][valid[i]]                          # [nv][2]
                    jj = np.nonzero(valid[i])[0]
                    w = qq0[i][None, None, :] * freq[:, jj[:, None], g].transpose(1, 2, 0)   # [nv][2][K]
                    cum = np.cumsum(w, -1); thr = (cum / cum[..., -1:])[..., :-1].reshape(nd, K - 1)
                    expc = (w / w.sum(-1, keepdims=True)).reshape(nd, K).sum(0)
                    cnts = {}
                    for d in range(-W, W + 1):
                        x = U[pos + 2 * d: pos + 2 * d + nd]
                        z = (x[:, None] > thr).sum(1)
                        cnts[d] = np.bincount(z, minlength=K).astype(float)
                    assert (cnts[0] == qn[i]).all(), (i, cnts[0], qn[i])
                    major = int(np.argmax(expc))
                    for d in range(-W, W + 1):
                        if d == 0: continue
                        st = pos + 2 * d + nd
                        ct = dirich_consume(U, st, cnts[d] + alpha)
                        stats["n"] += 1
                        stats["cs"].append((ct - 2 * K) // 2 if True else 0)
                        stats["mm_prev"] += ct != dirich_consume(U, st, prev_qqnum[i] + alpha)
                        stats["mm_nb"] += ct != dirich_consume(U, st, cnts[0] + alpha)
                        stats["mm_exp"] += ct != dirich_consume(U, st, expc + alpha)
                        # exact minor counts, major = total - minors but with +-30 error simulated by neighbour's major
                        sh = cnts[d].copy(); sh[major] = cnts[0][major]
                        stats["mm_sparse"] += ct != dirich_consume(U, st, sh + alpha)
                pos += nd + cons
            assert pos == used, (pos, used)
            n = stats["n"]; cs = np.array(stats["cs"])
            qs = np.sort(qq0, 1)[:, ::-1].mean(0)
            print(f"iter {it}: alpha={alpha:.4f} mean sorted qq={np.round(qs,4)} mean rejects/indiv={cs.mean():.2f} sd={cs.std():.2f}")
            print(f"   mismatch of consumption  prev-iter shapes {stats['mm_prev']/n:.4f}  neighbour-candidate shapes {stats['mm_nb']/n:.4f}"
                  f"  expected counts {stats['mm_exp']/n:.4f}  exact minors+approx major {stats['mm_sparse']/n:.4f}  (n={n})", flush=True)
        prev_qqnum = o.qqnum().copy()
        o.update_alpha(); o.cal_lkh()

if __name__ == "__main__":
    main()
