"""Parsing helpers for the tests/golden/*.golden trajectory files (format: oracle/dump_fmt.h)."""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, GOLDEN)
import make_golden  # noqa: E402  (case table + data generator; running it needs /root/reference, importing does not)

CASES = make_golden.CASES
HEXF = re.compile(r"-?0x[0-9a-f.]+p[+-]\d+|-?inf|-?nan")


def case_args(name):
    N, L, K, A, miss, u, b, t, c, e, y, r, j, seeds, mode, pf, detail, commit = CASES[name]
    return dict(N=N, L=L, K=K, A=A, miss=miss, u=u, b=b, t=t, c=c, e=e, y=y, r=r, j=j, seeds=seeds, mode=mode, pf=pf,
                detail=detail, commit=commit)


def case_text(name, tmpdir):
    """Path of the text data file of a case (committed for the small ones, regenerated otherwise)."""
    from instruct_amd import synth
    cfg = case_args(name)
    if cfg["commit"]:
        return os.path.join(GOLDEN, name + ".txt")
    base = {"c1_e0": "c1", "c1_y0": "c1", "c1_mode1": "c1", "c1_c2": "c1", "m4_c1": "c1", "m4_c1_e0": "c1", "m4_c1_miss": "c1_miss", "m3_c1": "c1", "m3_c1_miss": "c1_miss", "m5_c1": "c1", "m5_c1_miss": "c1_miss", "m0_c1": "c1", "m0_c1_miss": "c1_miss", "m0_c1_a3": "c1_a3"}.get(name, name)
    if os.path.exists(os.path.join(GOLDEN, base + ".txt")):
        return os.path.join(GOLDEN, base + ".txt")
    path = os.path.join(str(tmpdir), name + ".txt")
    synth.write_text_diploid(path, make_golden.data_for(name))
    return path


def case_data(name):
    """(geno, allelenum, missindx) of a case, coded as the reference reader codes it."""
    from instruct_amd import synth
    return synth.code_diploid(make_golden.data_for(name))


def dump_cmd_args(name, txt, out):
    c = case_args(name)
    return [txt, out] + [str(x) for x in (c["K"], c["N"], c["L"], c["u"], c["b"], c["t"], c["c"], c["e"], c["y"], c["r"], c["j"],
                                            c["seeds"][0], c["seeds"][1], c["seeds"][2], c["mode"], c["pf"], c["detail"])]


def fields(line):
    """key=value tokens of a line (seeds=a b c kept together)."""
    out = {}
    m = re.search(r"seeds=(\d+ \d+ \d+)", line)
    if m:
        out["seeds"] = tuple(int(x) for x in m.group(1).split())
    for k, v in re.findall(r"(\w+)=(\S+)", line):
        if k != "seeds":
            out[k] = v
    return out


def floats(line):
    return [float.fromhex(x) if x.startswith(("0x", "-0x")) else float(x) for x in HEXF.findall(line)]


def parse(path):
    """-> list of (tag tuple, raw line).  tag e.g. ('it', 3, 'ZQ') or ('chain', 'qq', 4)."""
    recs = []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if not line or line.startswith("#"):
                continue
            recs.append(line)
    return recs
