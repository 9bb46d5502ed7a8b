"""The constants the matrix-core kernels run on (csrc/poseidon_mx.cuh, csrc/ntt_mx.cuh), pinned on the CPU: the library's
host-side tables against independent Python derivations, and the integer model of the matrix-core NTT (digits, bias,
pass structure, LDS swizzle).  No GPU needed."""
import ctypes as C
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
P = 0xFFFFFFFF00000001
MDS_C = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]


def _lib():
    import proof_protocol_decoder_amd as pkg
    return pkg.lib()


def _round_constants():
    txt = open(os.path.join(ROOT, "proof_protocol_decoder_amd", "csrc", "poseidon_rc.inc")).read()
    rc = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{16})ULL", txt)]
    assert len(rc) == 360
    return rc


def test_poseidon_mx_c_operand_table():
    """[round][ib][2g + h][reg] = 128 * rowsum(word ib + 4g) + byte 4h + reg of the NEXT round's constant"""
    L = _lib()
    L.bp_debug_poseidon_mx_cin.argtypes = [C.POINTER(C.c_uint32)]
    buf = (C.c_uint32 * (30 * 4 * 24))()
    assert L.bp_debug_poseidon_mx_cin(buf) == 0
    got = np.frombuffer(buf, dtype=np.uint32).reshape(30, 4, 3, 2, 4)
    rc = _round_constants()
    for rnd in range(30):
        for ib in range(4):
            for g in range(3):
                wo = ib + 4 * g
                rowsum = sum(MDS_C) + (8 if wo == 0 else 0)
                for h in range(2):
                    for reg in range(4):
                        want = 128 * rowsum
                        if rnd < 29:
                            want += (rc[(rnd + 1) * 12 + wo] >> (8 * (4 * h + reg))) & 0xFF
                        assert int(got[rnd, ib, g, h, reg]) == want
    # the accumulator bound the recombination relies on: plane sums stay below 2^17
    assert 264 * 255 + 255 < 1 << 17


def test_mds_on_byte_planes_equals_the_field_mds():
    """the identity the kernel rests on, in integers: sum_p 2^(8p) * (M x plane_p(state)) = M x state (mod p), with the
    bytes taken as signed x - 128 and the 128 * rowsum correction"""
    rng = np.random.default_rng(5)
    M = [[MDS_C[(k - i) % 12] + (8 if i == 0 and k == 0 else 0) for k in range(12)] for i in range(12)]
    for _ in range(50):
        s = [int(x) for x in rng.integers(0, 1 << 64, 12, dtype=np.uint64)]
        want = [sum(M[i][k] * s[k] for k in range(12)) % P for i in range(12)]
        got = []
        for i in range(12):
            planes = [sum(M[i][k] * (((s[k] >> (8 * p)) & 0xFF) - 128) for k in range(12)) + 128 * sum(M[i])
                      for p in range(8)]
            assert all(0 <= a < (1 << 17) for a in planes)
            got.append(sum(a << (8 * p) for p, a in enumerate(planes)) % P)
        assert got == want


def test_ntt_mx_integer_model():
    import ntt_mx_model
    ntt_mx_model.main()            # DFT16 on byte planes, three-pass DIF / DIT pipelines, accumulator bounds
    ntt_mx_model.check_swizzle()   # every LDS access pattern of the passes is bank-conflict free


def test_ntt_mx_tables_match_the_model():
    """The A / C operand images and twiddle tables the device kernels load (built on the host by
    mxn::build_tables), against the integer model's matrix with the kernels' lane / row / slot assignment."""
    import ntt_mx_model as m
    L = _lib()
    L.bp_debug_ntt_mx_tables.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    br4 = lambda k: m.bitrev(k, 4)
    for kind in (0, 1):
        for inverse in (0, 1):
            a = (C.c_uint8 * 16384)()
            c = (C.c_int32 * 128)()
            tw256 = (C.c_uint64 * 4096)()
            tw16 = (C.c_uint64 * 256)()
            assert L.bp_debug_ntt_mx_tables(kind, inverse, a, c, tw256, tw16) == 0
            a = np.frombuffer(a, dtype=np.int8).reshape(8, 2, 64, 16)
            c = np.frombuffer(c, dtype=np.int32).reshape(8, 4, 4)
            w4096 = m.root(12)
            if inverse:
                w4096 = pow(w4096, P - 2, P)
            A, Cc = m.dft16_matrix(pow(w4096, 256, P))     # A[(k, q)][(j, p)], C[(k, q)]
            idx = (lambda x: br4(x)) if kind else (lambda x: x)
            for rb in range(8):
                g, h = rb >> 1, rb & 1
                for ch in range(2):
                    for lane in range(64):
                        r, kb = lane & 15, lane >> 4
                        out = idx((r >> 2) + 4 * g)
                        q = 4 * h + (r & 3)
                        for b in range(16):
                            inp = idx(8 * ch + 2 * kb + (b >> 3))
                            assert int(a[rb, ch, lane, b]) == A[out * 8 + q][inp * 8 + (b & 7)]
                for ib in range(4):
                    for reg in range(4):
                        assert int(c[rb, ib, reg]) == Cc[idx(ib + 4 * g) * 8 + 4 * h + reg]
            t256 = np.frombuffer(tw256, dtype=np.uint64).reshape(16, 256)
            t16 = np.frombuffer(tw16, dtype=np.uint64).reshape(16, 16)
            w256 = pow(w4096, 16, P)
            for k in (0, 1, 5, 15):
                for i in (0, 1, 17, 255):
                    assert int(t256[k, i]) == pow(w4096, i * k, P)
                for i in (0, 3, 15):
                    assert int(t16[k, i]) == pow(w256, i * k, P)
