"""Builds the native pieces in-tree (no JIT cache): the HIP library for gfx950 and the C host shim.

    python -m instruct_amd.build            # libinstruct_hip.so + host/mcmc_hip.o
    python -m instruct_amd.build --oracle   # also oracle/liborc.so (+ oracle/_ref when /root/reference exists)
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
LIB_HIP = os.path.join(PKG, "libinstruct_hip.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, **kw)


def hipcc():
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        p = shutil.which(c)
        if p:
            return p
    raise RuntimeError("hipcc not found")


def build_hip(force=False):
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "instruct_hip.h")]
    if force or _newer(LIB_HIP, srcs):
        _run([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
              "-Wall", "-Wno-unused-function", "-pthread", "-Rpass-analysis=kernel-resource-usage", "-o", LIB_HIP, os.path.join(CSRC, "isg_hip.hip")])
    return LIB_HIP


def build_host(force=False):
    """Compiles the drop-in sampler object (plain C).  It is linked into the host program in place
    of the reference's mcmc.o; oracle/Makefile links it against the reference
    driver objects for the end-to-end test binary oracle/_ref/InStruct_hip."""
    src = os.path.join(HOST, "mcmc_hip.c")
    obj = os.path.join(HOST, "mcmc_hip.o")
    srcs = [src, os.path.join(HOST, "instruct_types.h"), os.path.join(ROOT, "include", "instruct_hip.h")]
    if force or _newer(obj, srcs):
        _run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-Wall", "-I", os.path.join(ROOT, "include"), "-c", src, "-o", obj])
    # the streaming reader: in place of the reference's data_interface.o (optional, independent of the sampler)
    rsrc = os.path.join(HOST, "data_interface_stream.c")
    robj = os.path.join(HOST, "data_interface_stream.o")
    if force or _newer(robj, [rsrc, os.path.join(HOST, "instruct_types.h")]):
        _run(["gcc", "-O2", "-fPIC", "-Wall", "-c", rsrc, "-o", robj])
    # the benchmark's synthetic tetraploid generator (instruct_amd/synth.py uses it when it is there)
    ssrc, slib = os.path.join(HOST, "synth_fast.c"), os.path.join(PKG, "libisg_synth.so")
    if force or _newer(slib, [ssrc]):
        _run(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-Wall", ssrc, "-o", slib])
    # the multi-GPU launcher (plain C, no GPU call of its own)
    msrc, mexe = os.path.join(HOST, "instruct_mgpu.c"), os.path.join(HOST, "instruct_mgpu")
    if force or _newer(mexe, [msrc]):
        _run(["gcc", "-O2", "-Wall", msrc, "-o", mexe, "-lm"])
    return obj


def build_oracle():
    _run(["make", "-C", os.path.join(ROOT, "oracle"), "all"])
    if os.path.isdir("/root/reference"):
        _run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])


def main(argv):
    build_hip("--force" in argv)
    build_host("--force" in argv)
    if "--oracle" in argv:
        build_oracle()


if __name__ == "__main__":
    main(sys.argv[1:])
