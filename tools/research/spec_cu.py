#!/usr/bin/env python3
"""Research prototype: classify every rgamma attempt as CERTAIN (same outcome for every shape in
[lambda - 6 sigma, lambda + 6 sigma] + alpha) or UNCERTAIN, and measure how many (individual, start candidate) pairs have a
Dirichlet consumption that is known without the exact cluster counts.  lambda, sigma = mean / sd of the cluster counts given qq, freq."""
import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import spec_proto as sp
from spec_proto import wh_tape, orc, synth
E = sp.E

def try_margin(U, pos, a):
    """one attempt at tape position pos with shape a: returns (kind, accepted, margin-ish) ; consumption is 2 always here (a != 1)"""
    if a < 1:
        u0, u1 = U[pos], U[pos + 1]
        br = u0 > E / (a + E)
        if br:
            r = -math.log((a + E) * (1 - u0) / (a * E)); lim = r ** (a - 1) if r > 0 else float('inf')
        else:
            x = (a + E) * u0 / E; r = x ** (1 / a); lim = math.exp(-r)
        return (1 if br else 2), (not (u1 > lim)), lim - u1
    c1 = a - 1; c2 = (a - 1 / (6 * a)) / c1; c3 = 2 / c1; c4 = c3 + 2; c5 = 1 / math.sqrt(a)
    u1, u2 = U[pos], U[pos + 1]
    if a > 2.5: u1 = u2 + c5 * (1 - 1.86 * u1)
    if u1 >= 1 or u1 <= 0: return 3, False, min(abs(u1), abs(u1 - 1))
    w = c2 * u2 / u1
    q = c4 - (c3 * u1 + w + 1 / w)
    if q >= 0: return 4, True, q
    g = 1 - (c3 * math.log(u1) - math.log(w) + w)
    return 5, g > 0, g

def classify(U, pos, alo, ahi):
    """-> (certain?, accepted at mid)"""
    amid = 0.5 * (alo + ahi)
    km, am, mm = try_margin(U, pos, amid)
    if alo == ahi: return True, am
    if (alo < 1) != (ahi < 1) or (alo <= 2.5) != (ahi <= 2.5) or alo == 1 or ahi == 1 or (alo < 1 and ahi >= 1): return False, am
    kl, al, ml = try_margin(U, pos, alo); kh, ah, mh = try_margin(U, pos, ahi)
    if al != ah or al != am: return False, am
    # same outcome at both ends and in the middle; ask for branch agreement for a<1 and a margin larger than its own variation
    if alo < 1 and not (kl == kh == km): return False, am
    if (kl == 3) != (kh == 3) or (kl == 3) != (km == 3): return False, am
    return True, am

def dirich_cu(U, pos, lo, hi, alpha):
    """walk the K gammas with interval shapes; returns (certain, consumption with mid decisions)"""
    p0 = pos; cert = True
    for m in range(len(lo)):
        while True:
            c, acc = classify(U, pos, lo[m] + alpha, hi[m] + alpha)
            cert &= c; pos += 2
            if acc: break
    return cert, pos - p0

def main():
    N, L, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    iters = [int(x) for x in sys.argv[4].split(",")]
    NS = float(sys.argv[5]) if len(sys.argv) > 5 else 6.0
    geno, an, mi = synth.make_diploid(N, L, K)
    N, L, P = geno.shape
    o = orc.OrcChain(geno, an, mi, K)
    o.setseeds(13, 4, 1972)
    o.chain_init(np.array([o.ran1() for _ in range(K)], dtype=np.float32))
    valid = o.valid().astype(bool); nval = valid.sum(1)
    for it in range(max(iters) + 1):
        o.update_P(); o.update_S_POP(); o.update_G()
        if it in iters:
            qq0 = o.qq().copy(); freq = o.freq().copy(); alpha = o.alpha(); seeds = o.seeds(); c0 = o.rng_count()
        o.update_ZQ(0)
        if it in iters:
            used = int(o.rng_count() - c0)
            U = wh_tape(seeds, used + 40000)
            qn = o.qqnum()
            pos = 0; ncand = 0; nunc = 0; nwrong = 0; nunc_true = 0; width = []
            rng = np.random.default_rng(1)
            sample = set(rng.choice(N, size=min(N, 400), replace=False).tolist())
            W = 8
            for i in range(N):
                nd = 2 * int(nval[i])
                cons = sp.dirich_consume(U, pos + nd, qn[i] + alpha)
                if i in sample and pos > 2 * W:
                    g = geno[i][valid[i]]; jj = np.nonzero(valid[i])[0]
                    w = qq0[i][None, None, :] * freq[:, jj[:, None], g].transpose(1, 2, 0)
                    pr = (w / w.sum(-1, keepdims=True)).reshape(nd, K)
                    lam = pr.sum(0); sg = np.sqrt((pr * (1 - pr)).sum(0))
                    lo = np.maximum(0, np.floor(lam - NS * sg - 1)); hi = np.minimum(nd, np.ceil(lam + NS * sg + 1))
                    cum = np.cumsum(w, -1); thr = (cum / cum[..., -1:])[..., :-1].reshape(nd, K - 1)
                    width.append((hi - lo))
                    for d in range(-W, W + 1):
                        st = pos + 2 * d
                        z = (U[st:st + nd][:, None] > thr).sum(1)
                        cn = np.bincount(z, minlength=K).astype(float)
                        assert (cn >= lo).all() and (cn <= hi).all(), (cn, lo, hi)
                        cert, ccu = dirich_cu(U, st + nd, lo, hi, alpha)
                        cex = sp.dirich_consume(U, st + nd, cn + alpha)
                        ncand += 1; nunc += (not cert)
                        if d == 0: nunc_true += (not cert)
                        if cert and ccu != cex: nwrong += 1
                pos += nd + cons
            print(f"iter {it}: alpha={alpha:.3f} qq sorted={np.round(np.sort(qq0,1)[:,::-1].mean(0),3)} interval +-{NS} sigma, mean widths {np.round(np.mean(width,0),1)}:"
                  f" uncertain candidates {nunc/ncand:.4f} (on the true path {nunc_true/len(width):.4f}), certain-but-wrong {nwrong} of {ncand}", flush=True)
        o.update_alpha(); o.cal_lkh()
if __name__ == "__main__":
    main()
