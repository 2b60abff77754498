"""GPU: a long replay chain at config 3 in the four update_ZQ forms -- interval resolver (+ update_P on the device), block resolver in one
launch, one launch per block, round 1's chain kernels (the last three with update_P's host loop) -- must end in the same state (stream
position, log-likelihood, Z, qq).  usage: python tools/gpu_modes_agree.py [iters]"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np
    from instruct_amd import capi, synth
    iters = int(sys.argv[2])
    geno, an, mi = synth.make_diploid(10000, 5000, 5)
    h = capi.HipChain(geno, an, mi, 5)
    h.setseeds(13, 4, 1972)
    h.chain_init(np.array([h.ran1() for _ in range(5)], dtype=np.float32))
    h.run(iters)
    print(json.dumps({"seeds": list(h.seeds()), "totallkh": h.totallkh(), "z": hashlib.sha1(h.z().tobytes()).hexdigest(),
                      "qq": hashlib.sha1(h.qq().tobytes()).hexdigest(), "fallbacks": h.zq_fallbacks(), "resolve": h.zq_resolve_stats(),
                      "interval": h.zq_spec_stats(), "update_P": h.p_device_stats()}))
    sys.exit(0)
iters = sys.argv[1] if len(sys.argv) > 1 else "300"
res = {}
OLD = {"INSTRUCT_ZQ_SPEC_RESOLVE": "0", "INSTRUCT_P_DEVICE": "0"}
for name, env in (("interval resolver", {}), ("one launch", OLD), ("launch per block", dict(OLD, INSTRUCT_ZQ_RESOLVE_PERSIST="0")), ("chain kernels", dict(OLD, INSTRUCT_ZQ_RESOLVE="0"))):
    out = subprocess.check_output([sys.executable, os.path.abspath(__file__), "--child", iters], env=dict(os.environ, **env))
    res[name] = json.loads(out.decode().strip().splitlines()[-1])
    print(name, res[name], flush=True)
ref = res["chain kernels"]
for name, r in res.items():
    assert all(r[k] == ref[k] for k in ("seeds", "totallkh", "z", "qq")), name
    assert r["fallbacks"] == 0, name
print("MODES AGREE after", iters, "iterations")
