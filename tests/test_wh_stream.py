"""CPU: the product's random-access Wichmann-Hill stream (instruct_amd/csrc/isg_wh.h) -- the integer and
floating-point shortcuts used on the device are exhaustively equal to the plain formulas of random.c:19-47."""
import ctypes as C

import numpy as np

from instruct_amd import capi


def test_selftest_exhaustive_lcg_division_and_skip_ahead():
    lib = capi.load()
    assert lib.isg_selftest() == 0, lib.isg_last_error().decode()


def test_keyed_layout_is_the_documented_one():
    # pure arithmetic of include/instruct_hip.h "keyed layout" (checked against the oracle's own copy)
    import orc
    geno = np.zeros((7, 13, 2), dtype=np.int32)
    o = orc.OrcChain(geno, np.full(13, 3, dtype=np.int32), np.zeros((7, 13), dtype=np.int32), 4)
    SP, SZ, ZI0, B0, offS, offG, offZ, offA, BLK = o.keyed_layout()
    N, L, P, K, A = 7, 13, 2, 4, 3
    assert SP == 16 * A + 16 and SZ == P * L + 16 * K + 16 and ZI0 == 1 + 2 * N and B0 == ZI0 + N * SZ
    assert offS == K * L * SP and offG == offS + 4 * K and offZ == offG + 2 * N and offA == offZ + N * SZ and BLK == offA + 4
