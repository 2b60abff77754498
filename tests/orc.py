"""ctypes binding of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORC_DIR, "liborc.so")

MATH_LIBM, MATH_ISG = 0, 1
ACC_SEQ, ACC_EXACT = 0, 1
SCHED_REPLAY, SCHED_KEYED = 0, 1


class OrcParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "L", "P", "K", "Amax", "mode", "type_freq", "back_refl", "print_freq",
                                         "nstep_check_empty_cluster", "math", "accum", "sched")]


class OrcResult(C.Structure):
    _fields_ = [("steps", C.c_long), ("step", C.c_long), ("flag_empty_cluster", C.c_int), ("totallkh", C.c_double),
                ("totallkh2", C.c_double)] + [(n, C.POINTER(C.c_double)) for n in
                                              ("indvlkh", "self_rates", "self_rates2", "qq", "qq2", "gen", "gen2", "freq", "freq2")]


_lib = None


def build():
    subprocess.check_call(["make", "-C", ORC_DIR, "all"], stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        lib.orc_create.restype = C.c_void_p
        lib.orc_create.argtypes = [C.c_void_p] * 4
        for f in ("orc_z", "orc_freq", "orc_qq", "orc_qqnum", "orc_generation", "orc_self_rates", "orc_state", "orc_indvlkh",
                  "orc_valid"):
            getattr(lib, f).restype = C.c_void_p
            getattr(lib, f).argtypes = [C.c_void_p]
        for f in ("orc_alpha", "orc_totallkh", "orc_ran1"):
            getattr(lib, f).restype = C.c_double
            getattr(lib, f).argtypes = [C.c_void_p]
        lib.orc_set_alpha.argtypes = [C.c_void_p, C.c_double]
        lib.orc_set_seeds.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long]
        lib.orc_rng_count.restype = C.c_uint64
        lib.orc_rng_count.argtypes = [C.c_void_p]
        lib.orc_rgamma.restype = C.c_double
        lib.orc_rgamma.argtypes = [C.c_void_p, C.c_double]
        lib.orc_rgeom.argtypes = [C.c_void_p, C.c_double]
        lib.orc_rnormal.restype = C.c_double
        lib.orc_rnormal.argtypes = [C.c_void_p, C.c_double, C.c_double]
        lib.orc_rdirich.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double]
        lib.orc_disc_unif.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.orc_genofreq.restype = C.c_double
        lib.orc_genofreq.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
        lib.orc_gelman_rubin.restype = C.c_double
        lib.orc_gelman_rubin.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.orc_run_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        for f in ("orc_destroy", "orc_update_P", "orc_update_S_POP", "orc_update_F_POP", "orc_update_S_IND", "orc_update_F_IND", "orc_update_G", "orc_update_alpha", "orc_cal_lkh",
                  "orc_iteration"):
            getattr(lib, f).argtypes = [C.c_void_p]
        lib.orc_update_ZQ.argtypes = [C.c_void_p, C.c_int]
        lib.orc_update_Z.argtypes = [C.c_void_p, C.c_int]
        lib.orc_chain_init.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_get_seeds.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_count_alleles.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_keyed_get_layout.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_error.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _view(addr, shape, dtype):
    n = int(np.prod(shape))
    buf = (C.c_byte * (n * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class OrcChain:
    def __init__(self, geno, allelenum, missindx, K, mode=2, type_freq=1, back_refl=1, print_freq=0, nstep_check=5,
                 math=MATH_LIBM, accum=ACC_SEQ, sched=SCHED_REPLAY):
        self.lib = load()
        geno = np.ascontiguousarray(geno, dtype=np.int32)
        self.N, self.L, self.P = geno.shape
        self.K = K
        self.allelenum = np.ascontiguousarray(allelenum, dtype=np.int32)
        self.Amax = int(self.allelenum.max())
        missindx = np.ascontiguousarray(missindx, dtype=np.int32)
        self.p = OrcParams(self.N, self.L, self.P, K, self.Amax, mode, type_freq, back_refl, print_freq, nstep_check, math, accum, sched)
        self.h = self.lib.orc_create(C.addressof(self.p), _ptr(self.allelenum), _ptr(geno), _ptr(missindx))
        self.mode = mode

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_destroy(self.h)
            self.h = None

    def setseeds(self, a, b, c):
        self.lib.orc_set_seeds(self.h, a, b, c)

    def seeds(self):
        s = (C.c_long * 3)()
        self.lib.orc_get_seeds(self.h, s)
        return tuple(s)

    def ran1(self):
        return self.lib.orc_ran1(self.h)

    def rng_count(self):
        return self.lib.orc_rng_count(self.h)

    def chain_init(self, initd):
        v = np.ascontiguousarray(initd, dtype=np.float32)
        self.lib.orc_chain_init(self.h, _ptr(v))

    def update_P(self): self.lib.orc_update_P(self.h)
    def update_S_POP(self):
        (self.lib.orc_update_F_POP if self.mode == 4 else self.lib.orc_update_S_POP)(self.h)
    def update_S_IND(self):
        (self.lib.orc_update_F_IND if self.mode == 5 else self.lib.orc_update_S_IND)(self.h)
    def update_G(self): self.lib.orc_update_G(self.h)
    def update_ZQ(self, init_flag=0): self.lib.orc_update_ZQ(self.h, init_flag)
    def update_Z(self, init_flag=0): self.lib.orc_update_Z(self.h, init_flag)
    def update_alpha(self): self.lib.orc_update_alpha(self.h)
    def cal_lkh(self): self.lib.orc_cal_lkh(self.h)
    def iteration(self): self.lib.orc_iteration(self.h)

    def z(self):
        z = _view(self.lib.orc_z(self.h), (self.N, self.L, self.P), np.int32).copy()
        z[self.valid() == 0] = -1
        return z

    def z_raw(self): return _view(self.lib.orc_z(self.h), (self.N, self.L, self.P), np.int32)
    def valid(self): return _view(self.lib.orc_valid(self.h), (self.N, self.L), np.int32)
    def freq(self): return _view(self.lib.orc_freq(self.h), (self.K, self.L, self.Amax), np.float64)
    def qq(self): return _view(self.lib.orc_qq(self.h), (self.N, self.K), np.float64)
    def qqnum(self): return _view(self.lib.orc_qqnum(self.h), (self.N, self.K), np.float64)
    def generation(self): return _view(self.lib.orc_generation(self.h), (self.N,), np.int32)
    def self_rates(self): return _view(self.lib.orc_self_rates(self.h), (self.N if self.mode in (3, 5) else self.K,), np.float64)
    def state(self): return _view(self.lib.orc_state(self.h), (self.K,), np.int32)
    def indvlkh(self): return _view(self.lib.orc_indvlkh(self.h), (self.N,), np.float64)
    def alpha(self): return self.lib.orc_alpha(self.h)
    def set_alpha(self, a): self.lib.orc_set_alpha(self.h, a)
    def totallkh(self): return self.lib.orc_totallkh(self.h)
    def error(self): return self.lib.orc_error(self.h)

    def count_alleles(self):
        out = np.zeros((self.K, self.L, self.Amax), dtype=np.int32)
        self.lib.orc_count_alleles(self.h, _ptr(out))
        return out

    def keyed_layout(self):
        out = (C.c_uint64 * 9)()
        self.lib.orc_keyed_get_layout(self.h, out)
        return tuple(out)

    def run_chain(self, initd, update, burnin, thinning, ckrep):
        v = np.ascontiguousarray(initd, dtype=np.float32)
        convg = np.zeros(max(ckrep, 1), dtype=np.float64)
        res = OrcResult()
        self.lib.orc_run_chain(self.h, _ptr(v), update, burnin, thinning, ckrep, _ptr(convg), C.addressof(res))
        out = {"steps": res.steps, "step": res.step, "flag_empty_cluster": res.flag_empty_cluster,
               "totallkh": res.totallkh, "totallkh2": res.totallkh2, "convg": convg[:ckrep].copy()}
        shapes = {"indvlkh": (self.N,), "self_rates": (self.K,), "self_rates2": (self.K,), "qq": (self.N, self.K),
                  "qq2": (self.N, self.K), "gen": (self.N,), "gen2": (self.N,), "freq": (self.K, self.L, self.Amax),
                  "freq2": (self.K, self.L, self.Amax)}
        if res.step > 0 or res.steps >= 0:
            for k, shp in shapes.items():
                ptr = getattr(res, k)
                if ptr:
                    out[k] = np.ctypeslib.as_array(ptr, shape=(int(np.prod(shp)),)).reshape(shp).copy()
        self.lib.orc_result_free(C.addressof(res))
        return out


def fnv_i32(a):
    lib = load()
    lib.orc_fnv_i32.restype = C.c_uint64
    lib.orc_fnv_i32.argtypes = [C.c_void_p, C.c_long]
    a = np.ascontiguousarray(a, dtype=np.int32)
    return "%016x" % lib.orc_fnv_i32(_ptr(a), a.size)


def fnv_f64(a):
    lib = load()
    lib.orc_fnv_f64.restype = C.c_uint64
    lib.orc_fnv_f64.argtypes = [C.c_void_p, C.c_long]
    a = np.ascontiguousarray(a, dtype=np.float64)
    return "%016x" % lib.orc_fnv_f64(_ptr(a), a.size)


def gelman_rubin(vec, numchains, totrep):
    v = np.ascontiguousarray(vec, dtype=np.float64)
    return load().orc_gelman_rubin(_ptr(v), numchains, totrep)
