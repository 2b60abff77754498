"""Scratch driver: first GPU shake-out of the HIP path against the canonical oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from instruct_amd import synth, capi
import orc

def compare(tag, h, o, fields):
    ok = True
    for f in fields:
        a = getattr(h, f)(); b = getattr(o, f)()
        if isinstance(a, np.ndarray):
            same = np.array_equal(a, np.asarray(b))
            if not same:
                d = np.argwhere(a != np.asarray(b))
                print(f"  MISMATCH {tag} {f}: {len(d)} diffs, first {d[:3].tolist()} hip={a[tuple(d[0])]!r} orc={np.asarray(b)[tuple(d[0])]!r}")
        else:
            same = (a == b)
            if not same: print(f"  MISMATCH {tag} {f}: hip={a!r} orc={b!r}")
        ok &= bool(same)
    print(tag, "OK" if ok else "FAIL", flush=True)
    return ok

def run(N, L, K, miss, nall, sched, iters, mode=2, y=1, e=1):
    raw = synth.raw_alleles(N, L, K, 2, nall, miss, 7)
    geno, an, mi = synth.code_diploid(raw)
    h = capi.HipChain(geno, an, mi, K, mode=mode, type_freq=y, back_refl=e, rng_sched=sched)
    o = orc.OrcChain(geno, an, mi, K, mode=mode, type_freq=y, back_refl=e, math=1, accum=1, sched=sched)
    h.setseeds(13, 4, 1972); o.setseeds(13, 4, 1972)
    initd = np.array([h.ran1() for _ in range(K)], dtype=np.float32)
    initd2 = np.array([o.ran1() for _ in range(K)], dtype=np.float32)
    assert (initd == initd2).all()
    h.chain_init(initd); o.chain_init(initd)
    ok = compare(f"[{N}x{L} K{K} s{sched}] init", h, o, ["z", "qq", "qqnum", "generation", "alpha", "seeds"] if sched == 0 else ["z", "qq", "qqnum", "generation", "alpha"])
    for it in range(iters):
        h.update_P(); o.update_P()
        ok &= compare(f" it{it} P", h, o, ["count_alleles", "freq"] + (["seeds"] if sched == 0 else []))
        if mode == 2:
            h.update_S_POP(); o.update_S_POP()
            ok &= compare(f" it{it} S", h, o, ["self_rates"] + (["seeds"] if sched == 0 else []))
            h.update_G(); o.update_G()
            ok &= compare(f" it{it} G", h, o, ["generation"] + (["seeds"] if sched == 0 else []))
        h.update_ZQ(0); o.update_ZQ(0)
        ok &= compare(f" it{it} ZQ", h, o, ["z", "qq", "qqnum"] + (["seeds"] if sched == 0 else []))
        h.update_alpha(); o.update_alpha()
        ok &= compare(f" it{it} A", h, o, ["alpha"] + (["seeds"] if sched == 0 else []))
        h.cal_lkh(); o.cal_lkh()
        ok &= compare(f" it{it} L", h, o, ["indvlkh", "totallkh"])
        if sched == 1:
            # keyed schedule: oracle advances its iteration counter only inside orc_iteration
            o.lib.orc_iter_advance.argtypes = [__import__('ctypes').c_void_p]; o.lib.orc_iter_advance(o.h)
            h.lib.isg_iter_advance.argtypes = [__import__('ctypes').c_void_p]; h.lib.isg_iter_advance(h.h)
        if not ok: break
    return ok

if __name__ == "__main__":
    allok = True
    allok &= run(50, 100, 3, 0.0, 2, 0, 3)
    allok &= run(50, 100, 3, 0.05, 3, 0, 3)
    allok &= run(50, 100, 3, 0.05, 3, 1, 3)
    allok &= run(300, 1500, 5, 0.02, 2, 0, 2)
    allok &= run(300, 1500, 5, 0.02, 2, 1, 2)
    allok &= run(64, 200, 4, 0.0, 2, 0, 2, mode=1)
    allok &= run(64, 200, 4, 0.03, 2, 0, 2, y=0, e=0)
    print("ALL OK" if allok else "SOME FAILED")
    sys.exit(0 if allok else 1)
