#!/usr/bin/env python3
"""Research: the interval resolver's lenient walk under host emulation on the REAL first sweep of config 3 (oracle state)."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc, test_walk_emul as twe
sys.path.insert(0, os.path.join(ROOT, 'tools', 'research'))
import spec_proto as sp
from instruct_amd import synth
N, L, K = (int(x) for x in sys.argv[1:4])
seg = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
geno, an, mi = synth.make_diploid(N, L, K)
N, L, P = geno.shape
o = orc.OrcChain(geno, an, mi, K)
o.setseeds(13, 4, 1972)
o.chain_init(np.array([o.ran1() for _ in range(K)], dtype=np.float32))
t0 = time.time()
o.update_P(); o.update_S_POP(); o.update_G()
print("oracle sweeps", round(time.time() - t0, 1), "s", flush=True)
qq = o.qq().copy(); freq = o.freq().copy(); alpha = o.alpha(); seeds = o.seeds()
valid = o.valid().astype(bool); nval = valid.sum(1)
lam = np.zeros((N, K)); var = np.zeros((N, K))
for i in range(N):
    jj = np.nonzero(valid[i])[0]
    g = geno[i][jj]
    w = qq[i][None, None, :] * freq[:, jj[:, None], g].transpose(1, 2, 0)
    p = (w / w.sum(-1, keepdims=True)).reshape(-1, K)
    lam[i] = p.sum(0); var[i] = (p * (1 - p)).sum(0)
h = 4.5 * np.sqrt(var) + 1.0 + 1e-5 * lam
lo = np.maximum(0, np.floor(lam - h)); hi = np.minimum(2 * nval[:, None], np.ceil(lam + h))
alo = (lo + alpha).astype(np.float32).reshape(-1); ahi = (hi + alpha).astype(np.float32).reshape(-1)
gam0 = (np.arange(N + 1) * K).astype(np.int32)
B = np.concatenate([[0], np.cumsum(2 * nval + 2 * K)]).astype(np.uint64)
gpos = np.zeros(N * K + 1, dtype=np.uint64)
for m in range(K):
    gpos[m:N * K:K] = B[:N] + (2 * nval).astype(np.uint64) + np.uint64(2 * m)
gpos[N * K] = B[N]
l = twe.lib()
l.wk_emul_spec_begin.restype = C.c_long
l.wk_ref_consumed.restype = C.c_uint
band = int(sys.argv[5]) if len(sys.argv) > 5 else 0
l.wk_emul_spec_round.restype = C.c_long
T = np.zeros(N + 1, dtype=np.uint64); out = np.zeros(8, dtype=np.uint64)
cap = 60 * N
pg = np.zeros(cap, dtype=np.int32); px = np.zeros(cap, dtype=np.uint64)
t0 = time.time()
n = l.wk_emul_spec_begin(gam0.ctypes, N, gpos.ctypes, alo.ctypes, ahi.ctypes, C.c_long(seeds[0]), C.c_long(seeds[1]), C.c_long(seeds[2]), C.c_double(1.0), C.c_double(5.0), seg, band,
                         200, 256, T.ctypes, pg.ctypes, px.ctypes, C.c_long(cap), out.ctypes)
print("lenient", round(time.time() - t0, 1), "s probes", n, "fail", out[0], out[1], "T[N]/N", float(T[N]) / N, flush=True)
used_max = int(B[N]) + 2 * int(T[N]) + 2 * 200 * 50 + 100000
U = sp.wh_tape(seeds, used_max)
thr = {}
def answer(n):
    pc = np.zeros(max(n, 1), dtype=np.uint8)
    for k in range(n):
        i = int(pg[k]); x = int(px[k])
        if i not in thr:
            jj = np.nonzero(valid[i])[0]; g = geno[i][jj]
            w = qq[i][None, None, :] * freq[:, jj[:, None], g].transpose(1, 2, 0)
            cum = np.cumsum(w, -1); thr[i] = (cum / cum[..., -1:])[..., :-1].reshape(-1, K - 1)
        nd = 2 * int(nval[i]); p = int(B[i]) + 2 * x
        z = (U[p:p + nd][:, None] > thr[i]).sum(1)
        cn = np.ascontiguousarray(np.bincount(z, minlength=K).astype(np.float64) + alpha)
        used = l.wk_ref_consumed(C.c_ulonglong(p + nd), cn.ctypes, K, C.c_long(seeds[0]), C.c_long(seeds[1]), C.c_long(seeds[2]))
        c = (used - 2 * K) // 2
        pc[k] = c if c <= 126 else 255
    return pc
total = 0
for rnd in range(60):
    total += n
    pc = answer(n)
    Tprev = T.copy()
    n = l.wk_emul_spec_round(gam0.ctypes, gpos.ctypes, alo.ctypes, ahi.ctypes, C.c_long(seeds[0]), C.c_long(seeds[1]), C.c_long(seeds[2]), pc.ctypes, C.c_long(n), band, 256, T.ctypes,
                             pg.ctypes, px.ctypes, C.c_long(cap), out.ctypes)
    same = int((T == Tprev).sum())
    print("round", rnd, "new probes", n, "fail", out[0], "T[N]", int(T[N]), "individuals whose offset stayed", same, flush=True)
    if n == 0 or out[0]:
        break
print("probes in all", total)
c0 = o.rng_count()
o.update_ZQ(0)
print("true total rejections", (int(o.rng_count() - c0) - int(B[N])) // 2, "strict walk's", int(T[N]))
