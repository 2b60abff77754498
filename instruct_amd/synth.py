"""Deterministic synthetic genotype generator.

Produces the *coded* genotype tensor the hot path reads -- the layout of
``SEQDATA.seqdata`` (int32 ``[N][L][P]``, allele codes ``0..A_j-1`` in order of
first appearance scanning individuals then copies, ``-9`` = missing; reference
``data_interface.c:489-569`` ``transform_data``) -- plus, for the small parity
cases, the whitespace-token text file the reference reader parses
(``data_interface.c:91-245``).

The random stream is a counter-based splitmix64 evaluated with numpy uint64
arithmetic, so the same arrays come out on any machine / numpy version.
"""
from __future__ import annotations

import numpy as np

MISSING = -9

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _uniform(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """U[0,1) doubles keyed by (seed, stream, idx)."""
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(seed) + np.uint64(stream) * np.uint64(0xD1B54A32D192ED03))
        bits = _splitmix64(idx.astype(np.uint64) ^ key)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def raw_alleles(N: int, L: int, K: int, ploidy: int = 2, n_alleles: int = 2,
                missing_frac: float = 0.0, seed: int = 20260101) -> np.ndarray:
    """Raw allele labels (1-based ints, ``MISSING`` for missing) of shape [N][L][P].

    * K true clusters, individual i belongs to cluster ``i % K``;
    * per (cluster, locus) allele frequencies: biallelic p ~ U(0.05, 0.95), or for
      ``n_alleles > 2`` a normalised vector of n exponentials (Dirichlet(1..1));
    * every copy drawn from its cluster's frequencies; for diploids the second copy
      is replaced by the first with probability 0.5 (inbreeding signal);
    * ``missing_frac`` of the (i, j) cells are whole-locus missing.
    """
    P = ploidy
    kk = np.arange(K * L, dtype=np.uint64)
    if n_alleles == 2:
        p = 0.05 + 0.9 * _uniform(seed, 1, kk).reshape(K, L)
        cum = np.stack([p, np.ones_like(p)], axis=-1)  # [K][L][2]
    else:
        e = -np.log1p(-_uniform(seed, 1, np.arange(K * L * n_alleles, dtype=np.uint64)))
        e = e.reshape(K, L, n_alleles)
        cum = np.cumsum(e / e.sum(-1, keepdims=True), axis=-1)
        cum[..., -1] = 1.0
    out = np.empty((N, L, P), dtype=np.int32)
    clus = np.arange(N) % K
    # chunk over individuals to bound memory
    step = max(1, (1 << 22) // max(1, L * P))
    for i0 in range(0, N, step):
        i1 = min(N, i0 + step)
        n = i1 - i0
        idx = (np.arange(i0, i1, dtype=np.uint64)[:, None, None] * np.uint64(L * P)
               + np.arange(L, dtype=np.uint64)[None, :, None] * np.uint64(P)
               + np.arange(P, dtype=np.uint64)[None, None, :])
        u = _uniform(seed, 2, idx)                       # [n][L][P]
        c = cum[clus[i0:i1]]                             # [n][L][A]
        a = (u[..., None] >= c[:, :, None, :]).sum(-1).astype(np.int32) + 1  # 1-based label
        if P == 2:
            idx2 = (np.arange(i0, i1, dtype=np.uint64)[:, None] * np.uint64(L)
                    + np.arange(L, dtype=np.uint64)[None, :])
            same = _uniform(seed, 3, idx2) < 0.5
            a[..., 1] = np.where(same, a[..., 0], a[..., 1])
        if missing_frac > 0.0:
            idx2 = (np.arange(i0, i1, dtype=np.uint64)[:, None] * np.uint64(L)
                    + np.arange(L, dtype=np.uint64)[None, :])
            miss = _uniform(seed, 4, idx2) < missing_frac
            a[miss] = MISSING
        out[i0:i1] = a
    return out


def code_diploid(raw: np.ndarray):
    """Restates ``transform_data`` + ``get_missing`` (data_interface.c:489-569, 812-846).

    Returns (geno int32 [N][L'][P] with codes in first-appearance order, allelenum int32 [L'],
    missindx int32 [N][L']) with monomorphic loci dropped.
    """
    N, L, P = raw.shape
    flat = raw.transpose(1, 0, 2).reshape(L, N * P)  # [L][i*P+k]
    keep, codes, anum = [], [], []
    for j in range(L):
        col = flat[j]
        valid = col != MISSING
        vals, first = np.unique(col[valid], return_index=True)
        if vals.size < 2:
            continue
        order = np.argsort(first, kind="stable")          # first-appearance order
        rank = np.empty(vals.size, dtype=np.int32)
        rank[order] = np.arange(vals.size, dtype=np.int32)
        coded = np.full(col.shape, MISSING, dtype=np.int32)
        coded[valid] = rank[np.searchsorted(vals, col[valid])]
        keep.append(j)
        codes.append(coded)
        anum.append(vals.size)
    geno = np.stack(codes, axis=0).reshape(len(keep), N, P).transpose(1, 0, 2).copy()
    allelenum = np.asarray(anum, dtype=np.int32)
    missindx = (geno == MISSING).any(-1).astype(np.int32)
    return np.ascontiguousarray(geno, dtype=np.int32), allelenum, missindx


def code_biallelic_fast(raw: np.ndarray):
    """Vectorised ``code_diploid`` for labels in {1,2,MISSING} (large benchmark inputs)."""
    N, L, P = raw.shape
    flat = raw.transpose(1, 0, 2).reshape(L, N * P)
    valid = flat != MISSING
    first_idx = np.argmax(valid, axis=1)
    first_val = flat[np.arange(L), first_idx]
    has1 = ((flat == 1) & valid).any(1)
    has2 = ((flat == 2) & valid).any(1)
    keep = has1 & has2
    coded = np.where(valid, (flat != first_val[:, None]).astype(np.int32), MISSING)[keep]
    Lk = int(keep.sum())
    geno = coded.reshape(Lk, N, P).transpose(1, 0, 2).copy()
    allelenum = np.full(Lk, 2, dtype=np.int32)
    missindx = (geno == MISSING).any(-1).astype(np.int32)
    return np.ascontiguousarray(geno, dtype=np.int32), allelenum, missindx


def write_text_diploid(path: str, raw: np.ndarray) -> None:
    """Reference input format ``-af 0 -lb 0 -a 0``: P lines per individual, L tokens per line."""
    N, L, P = raw.shape
    with open(path, "w") as f:
        for i in range(N):
            for k in range(P):
                f.write(" ".join(str(int(v)) for v in raw[i, :, k]))
                f.write("\n")


def read_text_diploid(path: str, ploidy: int = 2) -> np.ndarray:
    """Parse the ``-af 0 -lb 0 -a 0`` text format back into raw labels [N][L][P] (ints only)."""
    rows = []
    with open(path) as f:
        for line in f:
            t = line.split()
            if t:
                rows.append([int(x) for x in t])
    arr = np.asarray(rows, dtype=np.int32)
    N = arr.shape[0] // ploidy
    return arr.reshape(N, ploidy, arr.shape[1]).transpose(0, 2, 1).copy()


def write_text_polyploid(path: str, raw: np.ndarray) -> None:
    """Reference input format for ploidy 4 (one line per individual, P adjacent tokens per locus;
    ``data_interface.c:247-350``)."""
    N, L, P = raw.shape
    with open(path, "w") as f:
        for i in range(N):
            f.write(" ".join(str(int(v)) for v in raw[i].reshape(-1)))
            f.write("\n")


def code_tetraploid(raw: np.ndarray):
    """Restates ``transform_data2`` + ``get_missing_tetra`` (data_interface.c:571-669, 722-741).

    Returns (obs int32 [N][L][P]: the sorted DISTINCT allele codes of each (individual, locus), padded
    with -1; alleleid int32 [N][L]: how many there are (0 = missing); allelenum int32 [L]).  Codes are
    numbered per locus in order of first appearance scanning individuals, then copies; loci are not dropped.
    """
    N, L, P = raw.shape
    obs = np.full((N, L, P), -1, dtype=np.int32)
    alleleid = np.zeros((N, L), dtype=np.int32)
    allelenum = np.zeros(L, dtype=np.int32)
    for j in range(L):
        col = raw[:, j, :].reshape(-1)
        valid = col != MISSING
        vals, first = np.unique(col[valid], return_index=True)
        order = np.argsort(first, kind="stable")
        rank = np.empty(vals.size, dtype=np.int32)
        rank[order] = np.arange(vals.size, dtype=np.int32)
        allelenum[j] = vals.size
        coded = np.full(col.shape, -1, dtype=np.int32)
        if vals.size:
            coded[valid] = rank[np.searchsorted(vals, col[valid])]
        coded = coded.reshape(N, P)
        for i in range(N):
            d = np.unique(coded[i][coded[i] >= 0])
            alleleid[i, j] = d.size
            obs[i, j, :d.size] = d
    return obs, alleleid, allelenum


def code_tetraploid_fast(raw: np.ndarray, max_label: int = 8):
    """Vectorised :func:`code_tetraploid` for integer labels 1..max_label (the generator's output): same
    (obs, alleleid, allelenum), sized for the benchmark workloads (N L P ~ 1e9)."""
    N, L, P = raw.shape
    big = np.int64(N) * P
    first = np.full((max_label + 1, L), big, dtype=np.int64)  # first appearance scanning individuals, then copies
    step = max(1, (1 << 24) // max(1, L * P))
    for i0 in range(0, N, step):
        blk = raw[i0:i0 + step]
        n = blk.shape[0]
        pos = (np.arange(i0, i0 + n, dtype=np.int64)[:, None, None] * P + np.arange(P, dtype=np.int64)[None, None, :])
        for v in range(1, max_label + 1):
            m = np.where(blk == v, pos, big).min(axis=(0, 2))
            np.minimum(first[v], m, out=first[v])
    if ((raw != MISSING) & ((raw < 1) | (raw > max_label))).any():
        raise ValueError("labels outside 1..max_label")
    order = np.argsort(first[1:], axis=0, kind="stable")  # [label rank][L] -> label index
    code = np.empty((max_label, L), dtype=np.int32)
    np.put_along_axis(code, order, np.arange(max_label, dtype=np.int32)[:, None].repeat(L, 1), axis=0)
    allelenum = (first[1:] < big).sum(axis=0).astype(np.int32)
    obs = np.full((N, L, P), -1, dtype=np.int32)
    alleleid = np.zeros((N, L), dtype=np.int32)
    jj = np.arange(L)
    for i0 in range(0, N, step):
        blk = raw[i0:i0 + step]
        mask = np.zeros(blk.shape[:2], dtype=np.int32)
        for k in range(P):
            v = blk[:, :, k]
            ok = v != MISSING
            c = code[np.where(ok, v, 1) - 1, jj[None, :]]
            mask |= np.where(ok, 1 << c, 0)
        fill = np.zeros(blk.shape[:2], dtype=np.int32)
        o = obs[i0:i0 + step]
        for c in range(max_label):
            has = (mask >> c) & 1 == 1
            slot = np.where(has, fill, P)  # P: dropped
            for k in range(P):
                sel = slot == k
                o[:, :, k][sel] = c
            fill += has
        alleleid[i0:i0 + step] = fill
    return obs, alleleid, allelenum


def make_diploid(N, L, K, missing_frac=0.0, n_alleles=2, seed=20260101):
    raw = raw_alleles(N, L, K, 2, n_alleles, missing_frac, seed)
    if n_alleles == 2:
        return code_biallelic_fast(raw)
    return code_diploid(raw)


def _cum_alleles(K, L, n_alleles, seed):
    """cumulative allele frequencies [K][L][A] exactly as raw_alleles forms them"""
    kk = np.arange(K * L, dtype=np.uint64)
    if n_alleles == 2:
        p = 0.05 + 0.9 * _uniform(seed, 1, kk).reshape(K, L)
        return np.stack([p, np.ones_like(p)], axis=-1)
    e = -np.log1p(-_uniform(seed, 1, np.arange(K * L * n_alleles, dtype=np.uint64)))
    e = e.reshape(K, L, n_alleles)
    cum = np.cumsum(e / e.sum(-1, keepdims=True), axis=-1)
    cum[..., -1] = 1.0
    return cum


def make_tetraploid_fast(N, L, K, n_alleles=4, missing_frac=0.05, seed=20260105):
    """(obs, alleleid, allelenum) of code_tetraploid(raw_alleles(N, L, K, 4, n_alleles, missing_frac, seed)), built by
    instruct_amd/host/synth_fast.c (libisg_synth.so) in seconds at config 5's size; the numpy path if the library is not there."""
    import ctypes as C
    import os
    lib_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libisg_synth.so")
    if not os.path.exists(lib_path):
        return code_tetraploid_fast(raw_alleles(N, L, K, 4, n_alleles, missing_frac, seed))
    lib = C.CDLL(lib_path)
    lib.isg_synth_tetraploid.argtypes = [C.c_long, C.c_long, C.c_int, C.c_int, C.c_double, C.c_uint64] + [C.c_void_p] * 4
    cum = np.ascontiguousarray(_cum_alleles(K, L, n_alleles, seed), dtype=np.float64)
    obs = np.empty((N, L, 4), dtype=np.int32)
    alleleid = np.empty((N, L), dtype=np.int32)
    allelenum = np.empty(L, dtype=np.int32)
    rc = lib.isg_synth_tetraploid(N, L, K, n_alleles, missing_frac, seed, cum.ctypes.data, obs.ctypes.data, alleleid.ctypes.data, allelenum.ctypes.data)
    if rc != 0:
        raise RuntimeError("isg_synth_tetraploid failed")
    return obs, alleleid, allelenum
