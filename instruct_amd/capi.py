"""ctypes binding of the C ABI in include/instruct_hip.h (libinstruct_hip.so).

Host-side mirror of the reference's sampler interface for one chain: method names follow the
reference functions (mcmc.c) each entry point replaces.  There is no CPU fallback: constructing a
:class:`HipChain` without a usable HIP device raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libinstruct_hip.so")

SCHED_REPLAY, SCHED_KEYED = 0, 1

EXPORTS = [
    "isg_ctx_create", "isg_ctx_destroy", "isg_last_error", "isg_set_seeds", "isg_get_seeds", "isg_ran1",
    "isg_chain_init", "isg_update_P", "isg_update_S_POP", "isg_update_S_IND", "isg_update_Z", "isg_update_G", "isg_update_ZQ", "isg_update_alpha",
    "isg_cal_lkh", "isg_iteration", "isg_run", "isg_iter_advance", "isg_count_alleles", "isg_get_z", "isg_get_freq", "isg_get_qq",
    "isg_get_qqnum", "isg_get_generation", "isg_get_self_rates", "isg_get_state", "isg_get_indvlkh",
    "isg_get_alpha", "isg_get_totallkh", "isg_get_amax", "isg_set_z", "isg_set_freq", "isg_set_qq",
    "isg_set_generation", "isg_set_self_rates", "isg_set_alpha", "isg_keyed_layout", "isg_profile_enable",
    "isg_profile_count", "isg_profile_get", "isg_profile_reset", "isg_gelman_rubin", "isg_selftest",
    "isg_store_begin", "isg_store_step", "isg_store_fetch", "isg_zq_fallbacks", "isg_p_device_stats", "isg_zq_spec_stats", "isg_copy_bandwidth", "isg_zq_resolve_stats", "isg_zq_resolve_plan", "isg_gather_convg",
    "isg_ctx_create_poly", "isg_poly_update_geno", "isg_get_poly_geno", "isg_get_poly_gs", "isg_get_poly_table", "isg_get_poly_freq2",
]


class IsgConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "L", "P", "K", "mode", "type_freq", "back_refl", "rng_sched", "device")] + [
        ("reserved", C.c_int32 * 7)]


_lib = None


def load():
    """Loads libinstruct_hip.so (raises if it has not been built -- there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `python -m instruct_amd.build` (HIP extension is mandatory)")
        lib = C.CDLL(LIB_PATH)
        lib.isg_last_error.restype = C.c_char_p
        lib.isg_ran1.restype = C.c_double
        lib.isg_ran1.argtypes = [C.c_void_p]
        lib.isg_gelman_rubin.restype = C.c_double
        lib.isg_gelman_rubin.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.isg_set_seeds.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long]
        lib.isg_set_alpha.argtypes = [C.c_void_p, C.c_double]
        lib.isg_run.argtypes = [C.c_void_p, C.c_long]
        lib.isg_zq_fallbacks.restype = C.c_long
        lib.isg_zq_fallbacks.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class IsgError(RuntimeError):
    pass


class HipChain:
    """One MCMC chain on one MI355X (UPMCMC state device resident)."""

    def __init__(self, geno, allelenum, missindx, K, mode=2, type_freq=1, back_refl=1, rng_sched=SCHED_REPLAY, device=0):
        self.lib = load()
        geno = np.ascontiguousarray(geno, dtype=np.int32)
        self.N, self.L, self.P = geno.shape
        self.K = K
        allelenum = np.ascontiguousarray(allelenum, dtype=np.int32)
        missindx = np.ascontiguousarray(missindx, dtype=np.int32)
        cfg = IsgConfig(self.N, self.L, self.P, K, mode, type_freq, back_refl, rng_sched, device)
        h = C.c_void_p()
        self._chk(self.lib.isg_ctx_create(C.byref(cfg), _ptr(allelenum), _ptr(geno), _ptr(missindx), C.byref(h)))
        self.h = h
        a = C.c_int32()
        self.lib.isg_get_amax(self.h, C.byref(a))
        self.Amax = a.value
        self.mode = mode

    def _chk(self, rc):
        if rc != 0:
            raise IsgError(self.lib.isg_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.isg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- RNG (random.c:50-63)
    def setseeds(self, s1, s2, s3):
        self._chk(self.lib.isg_set_seeds(self.h, s1, s2, s3))

    def seeds(self):
        s = (C.c_long * 3)()
        self._chk(self.lib.isg_get_seeds(self.h, s))
        return tuple(s)

    def ran1(self):
        return self.lib.isg_ran1(self.h)

    # --- sweeps (mcmc.c)
    def chain_init(self, initd):
        v = np.ascontiguousarray(initd, dtype=np.float32)
        self._chk(self.lib.isg_chain_init(self.h, _ptr(v)))

    def update_P(self):
        self._chk(self.lib.isg_update_P(self.h))

    def update_S_POP(self):
        self._chk(self.lib.isg_update_S_POP(self.h))

    def update_S_IND(self):
        self._chk(self.lib.isg_update_S_IND(self.h))

    def update_Z(self, init_flag=0):
        """mode 0: whole individuals are assigned; zz is returned by generation()"""
        self._chk(self.lib.isg_update_Z(self.h, init_flag))

    def update_G(self):
        self._chk(self.lib.isg_update_G(self.h))

    def update_ZQ(self, init_flag=0):
        self._chk(self.lib.isg_update_ZQ(self.h, init_flag))

    def update_alpha(self):
        self._chk(self.lib.isg_update_alpha(self.h))

    def cal_lkh(self):
        self._chk(self.lib.isg_cal_lkh(self.h))

    def iteration(self):
        self._chk(self.lib.isg_iteration(self.h))

    def run(self, n):
        self._chk(self.lib.isg_run(self.h, n))

    # --- state
    def _get(self, fn, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self._chk(getattr(self.lib, fn)(self.h, _ptr(out)))
        return out

    def count_alleles(self):
        return self._get("isg_count_alleles", (self.K, self.L, self.Amax), np.int32)

    def z(self):
        return self._get("isg_get_z", (self.N, self.L, self.P), np.int32)

    def freq(self):
        return self._get("isg_get_freq", (self.K, self.L, self.Amax), np.float64)

    def qq(self):
        return self._get("isg_get_qq", (self.N, self.K), np.float64)

    def qqnum(self):
        return self._get("isg_get_qqnum", (self.N, self.K), np.float64)

    def generation(self):
        return self._get("isg_get_generation", (self.N,), np.int32)

    def self_rates(self):
        return self._get("isg_get_self_rates", (self.N if self.mode in (3, 5) else self.K,), np.float64)

    def state(self):
        return self._get("isg_get_state", (self.K,), np.int32)

    def indvlkh(self):
        return self._get("isg_get_indvlkh", (self.N,), np.float64)

    def alpha(self):
        v = C.c_double()
        self._chk(self.lib.isg_get_alpha(self.h, C.byref(v)))
        return v.value

    def totallkh(self):
        v = C.c_double()
        self._chk(self.lib.isg_get_totallkh(self.h, C.byref(v)))
        return v.value

    def set_z(self, z):
        self._chk(self.lib.isg_set_z(self.h, _ptr(np.ascontiguousarray(z, dtype=np.int32))))

    def set_freq(self, f):
        self._chk(self.lib.isg_set_freq(self.h, _ptr(np.ascontiguousarray(f, dtype=np.float64))))

    def set_qq(self, q):
        self._chk(self.lib.isg_set_qq(self.h, _ptr(np.ascontiguousarray(q, dtype=np.float64))))

    def set_generation(self, g):
        self._chk(self.lib.isg_set_generation(self.h, _ptr(np.ascontiguousarray(g, dtype=np.int32))))

    def set_self_rates(self, s):
        self._chk(self.lib.isg_set_self_rates(self.h, _ptr(np.ascontiguousarray(s, dtype=np.float64))))

    def set_alpha(self, a):
        self._chk(self.lib.isg_set_alpha(self.h, float(a)))

    def keyed_layout(self):
        out = (C.c_uint64 * 9)()
        self._chk(self.lib.isg_keyed_layout(self.h, out))
        return tuple(out)

    # --- CHAIN running means on the device (store_chn, mcmc.c:1320-1456)
    def store_begin(self, with_freq=False):
        self._chk(self.lib.isg_store_begin(self.h, 1 if with_freq else 0))

    def store_step(self):
        self._chk(self.lib.isg_store_step(self.h))

    def store_fetch(self, want=("qq", "qq2", "indvlkh", "gen", "gen2")):
        """dict of the requested running means (+ "steps"); freq / freq2 as [K][L][Amax]"""
        N, K, L = self.N, self.K, self.L
        shapes = {"qq": (N, K), "qq2": (N, K), "indvlkh": (N,), "gen": (N,), "gen2": (N,), "freq": (K, L, self.Amax), "freq2": (K, L, self.Amax)}
        out = {k: np.zeros(shapes[k], dtype=np.float64) for k in want}
        steps = C.c_long()
        args = [_ptr(out[k]) if k in out else None for k in ("qq", "qq2", "indvlkh", "gen", "gen2", "freq", "freq2")]
        self._chk(self.lib.isg_store_fetch(self.h, *args, C.byref(steps)))
        out["steps"] = steps.value
        return out

    def zq_fallbacks(self):
        """replay update_ZQ sweeps that were redone by the single-workgroup kernel (see include/instruct_hip.h)"""
        return self.lib.isg_zq_fallbacks(self.h)

    def zq_spec_stats(self):
        """replay update_ZQ by shape intervals (isg_spec_hip.inc): sweeps tried / settled / lost, probes and fail bits of the last one"""
        out = (C.c_long * 10)()
        self._chk(self.lib.isg_zq_spec_stats(self.h, out))
        return dict(zip(("tried", "settled", "lost", "probes", "fail_bits", "rounds", "table_bytes", "segments", "lost_fail_bits", "retried"), out))

    def p_device_stats(self):
        """replay update_P on the device (walk engine): sweeps done there / by the host loop, plan size, window statistics"""
        out = (C.c_long * 10)()
        self._chk(self.lib.isg_p_device_stats(self.h, out))
        return dict(zip(("device_sweeps", "host_sweeps", "segments", "blocks", "table_bytes", "sigma_x1000", "scale_x1000", "kwin_x1000", "retried", "failed_runs"), out))

    def gather_convg(self, rank, world, id_path, mine):
        """ncclAllGather of this chain's log-likelihood samples (RCCL on device buffers) -> [world * n]"""
        mine = np.ascontiguousarray(mine, dtype=np.float64)
        out = np.empty(world * mine.size, dtype=np.float64)
        self._chk(self.lib.isg_gather_convg(self.h, rank, world, str(id_path).encode(), _ptr(mine), mine.size, _ptr(out)))
        return out

    def zq_resolve_stats(self):
        out = (C.c_long * 8)()
        self._chk(self.lib.isg_zq_resolve_stats(self.h, out))
        return dict(zip(("blocks", "misses", "launches", "D", "units", "mu_x1000", "sigma_x1000", "exact_redone"), out))

    # --- profiling
    def profile(self, on=True):
        self.lib.isg_profile_enable(self.h, 1 if on else 0)

    def profile_reset(self):
        self.lib.isg_profile_reset(self.h)

    def profile_results(self):
        res = {}
        for i in range(self.lib.isg_profile_count(self.h)):
            name = C.create_string_buffer(64)
            ms, n = C.c_double(), C.c_long()
            self._chk(self.lib.isg_profile_get(self.h, i, name, 64, C.byref(ms), C.byref(n)))
            res[name.value.decode()] = (ms.value, n.value)
        return res


class HipPolyChain(HipChain):
    """One ploidy-4 (autotetraploid, ``-p 4 -ap 1``) chain on one MI355X: the sweeps of poly_geno.c:98-116.

    obs: int32 [N][L][4] sorted distinct allele codes of each (individual, locus); alleleid int32 [N][L]: how many
    (0 = missing) -- SEQDATA.seqdata / SEQDATA.alleleid as transform_data2 (data_interface.c:571-669) leaves them.
    """

    def __init__(self, obs, alleleid, allelenum, K, back_refl=1, rng_sched=SCHED_REPLAY, device=0, allo=False):
        """allo=True: allotetraploid (``-ap 0``): two subgenomes with their own allele frequencies (freq / freq2)"""
        self.lib = load()
        self.allo = bool(allo)
        obs = np.ascontiguousarray(obs, dtype=np.int32)
        self.N, self.L, self.P = obs.shape
        self.K = K
        allelenum = np.ascontiguousarray(allelenum, dtype=np.int32)
        alleleid = np.ascontiguousarray(alleleid, dtype=np.int32)
        cfg = IsgConfig(self.N, self.L, self.P, K, 2, 1, back_refl, rng_sched, device)
        cfg.reserved[0] = 1 if allo else 0
        h = C.c_void_p()
        self._chk(self.lib.isg_ctx_create_poly(C.byref(cfg), _ptr(allelenum), _ptr(obs), _ptr(alleleid), C.byref(h)))
        self.h = h
        a = C.c_int32()
        self.lib.isg_get_amax(self.h, C.byref(a))
        self.Amax = a.value
        self.mode = 2
        gs = C.c_int32()
        self.gcount = np.empty(self.L, dtype=np.int32)
        self._chk(self.lib.isg_get_poly_gs(self.h, C.byref(gs), _ptr(self.gcount)))
        self.GS = gs.value

    def update_geno(self):
        self._chk(self.lib.isg_poly_update_geno(self.h))

    def freq2(self):
        """second subgenome's allele frequencies [K][L][Amax] (UPMCMC.freq2, allotetraploid)"""
        return self._get("isg_get_poly_freq2", (self.K, self.L, self.Amax), np.float64)

    def geno(self):
        return self._get("isg_get_poly_geno", (self.N, self.L, 4), np.int32)

    def _table(self, which):
        out = np.empty((self.K, self.L, self.GS), dtype=np.float32)
        self._chk(self.lib.isg_get_poly_table(self.h, which, _ptr(out)))
        return out

    def exfreq(self):
        """log expected genotype frequencies without selfing, float [K][L][GS] (calc_exfreq_auto, poly_geno.c:1515)"""
        return self._table(0)

    def genofreq(self):
        """log genotype frequencies at the clusters' selfing rates (auto_genfreq, poly_geno.c:1803)"""
        return self._table(1)

    def packed(self, tab):
        """the used entries of a table in the reference's order (k, j, g < G_j)"""
        mask = np.arange(self.GS)[None, :] < self.gcount[:, None]
        return np.ascontiguousarray(tab[:, mask])


def copy_bandwidth(device=0, nbytes=1 << 30, reps=8):
    """device-to-device copy bandwidth in GB/s (read + written), 16-byte accesses (isg_copy_bandwidth)"""
    lib = load()
    lib.isg_copy_bandwidth.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    g = C.c_double(0)
    if lib.isg_copy_bandwidth(device, nbytes, reps, C.byref(g)) != 0:
        raise IsgError(lib.isg_last_error().decode())
    return g.value


def gelman_rubin(vec, numchains, totrep):
    v = np.ascontiguousarray(vec, dtype=np.float64)
    return load().isg_gelman_rubin(_ptr(v), numchains, totrep)
