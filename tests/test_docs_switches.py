"""Every INSTRUCT_* environment switch INTEGRATION.md documents is read somewhere in the sources (and the other way round for the
switches of the default paths): the documentation of the diagnostic switches cannot drift from the code unnoticed."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(paths):
    return "\n".join(open(p, errors="replace").read() for p in paths)


def test_documented_switches_exist_in_the_sources():
    doc = _read([os.path.join(ROOT, "INTEGRATION.md")])
    src = _read(glob.glob(os.path.join(ROOT, "instruct_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "instruct_amd", "host", "*.c")) +
                glob.glob(os.path.join(ROOT, "instruct_amd", "*.py")))
    documented = set(re.findall(r"INSTRUCT_[A-Z0-9_]+", doc))
    read = set(re.findall(r'getenv\("(INSTRUCT_[A-Z0-9_]+)"\)', src)) | set(re.findall(r'environ(?:\.get)?\(?\[?"(INSTRUCT_[A-Z0-9_]+)"', src))
    missing = sorted(documented - read)
    assert not missing, "documented in INTEGRATION.md but read nowhere: %s" % missing
    # the switches of this round's default paths are documented
    for name in ("INSTRUCT_P_DEVICE", "INSTRUCT_WALK_SEG", "INSTRUCT_WALK_K", "INSTRUCT_ZQ_SPEC_RESOLVE", "INSTRUCT_ZQ_SPEC_KSIG", "INSTRUCT_ZQ_SPEC_SEG",
                 "INSTRUCT_ZQ_SPEC_ROUNDS", "INSTRUCT_ZEXPECT_STRIP", "INSTRUCT_LL_INT", "INSTRUCT_LL_TABLES"):
        assert name in documented and name in read, name
