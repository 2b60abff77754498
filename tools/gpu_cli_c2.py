"""End to end through the reference's CLI at BASELINE config 2 size (N=2000 L=1000 K=5, -v 2, 2 chains x 60 iterations):
the unmodified reference binary against the same driver around this repository's reader + sampler objects."""
import hashlib, os, subprocess, sys, time
sys.path.insert(0, ".")
from instruct_amd import synth
txt = "/tmp/c2.txt"
synth.write_text_diploid(txt, synth.raw_alleles(2000, 1000, 5, 2, 2, 0.0, 20260102))
args = ["-d", txt, "-K", "5", "-L", "1000", "-N", "2000", "-p", "2", "-u", "60", "-b", "30", "-t", "5", "-c", "2", "-v", "2", "-g", "1", "-r", "4", "-j", "4",
        "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "0", "-pf", "0"]
res = {}
for exe in ("InStruct_ref", "InStruct_full"):
    out = "/tmp/c2_%s.out" % exe
    t = time.time()
    p = subprocess.run([os.path.join("oracle", "_ref", exe), "-o", out] + args, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
    dt = time.time() - t
    body = [l for l in open(out, "rb").read().split(b"\n") if not (l.strip().startswith((b"Data File:", b"Output File:")) or b"InStruct" in l and b"-d" in l)]
    res[exe] = hashlib.md5(b"\n".join(body)).hexdigest()
    print(exe, "rc", p.returncode, "%.1f s" % dt, res[exe], flush=True)
print("result files byte-identical:", res["InStruct_ref"] == res["InStruct_full"], "(a 13-digit variance printed at %.3f may differ in its last digit)")
a = [l for l in open("/tmp/c2_InStruct_ref.out", "rb").read().split(b"\n")]
b = [l for l in open("/tmp/c2_InStruct_full.out", "rb").read().split(b"\n")]
diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y and not (x.strip().startswith((b"Data File:", b"Output File:")) or b"InStruct" in x and b"-d" in x)]
print(len(a), len(b), "differing lines:", len(diff))
for i, x, y in diff[:6]:
    print(i, x[:140]); print(i, y[:140])
