"""The constants the matrix-core kernels run on (csrc/poseidon_mx.cuh, csrc/ntt_mx.cuh), pinned on the CPU: the library's
host-side tables against independent Python derivations, and the integer model of the matrix-core NTT (digits, bias,
pass structure, LDS swizzle).  No GPU needed."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

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


def test_partial_round_linear_forms_model():
    """groundwork for a cheaper partial-round form: the 22 S-box inputs as linear forms of the
    11 untouched words and the earlier S-box outputs reproduce the plain rounds"""
    import poseidon_partial_model
    poseidon_partial_model.main()


def test_ntt_mx_integer_model():
    import ntt_mx_model
    ntt_mx_model.main()            # DFT16 on byte planes, three-pass DIF / DIT pipelines, accumulator bounds
    ntt_mx_model.check_swizzle()   # every LDS access pattern of the passes is bank-conflict free


def test_ntt_mx_tables_match_the_model():
    """The A / C operand images and twiddle tables the device kernels load (built on the host by
    mxn::build_tables), against the integer model's matrix with the kernels' lane / row / slot assignment.
    Only K-chunk 0 is stored: the coefficient of input index + 8 is (-1)^(output index) times it, every tile row has
    an output index of parity a & 1, and odd rows run chunk 1 against the bytes XOR 0x7F (= 127 - x), which moves
    -127 * (digit sum) instead of +128 * (digit sum) into the row constant."""
    import ntt_mx_model as m
    L = _lib()
    L.bp_debug_ntt_mx_tables.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_int32),
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    for kind in (0, 1):
        in_index = (lambda kb, eps: 4 * eps + kb) if kind else (lambda kb, eps: 2 * kb + eps)
        out_index = (lambda ib, a: (a & 1) + 2 * ib + 8 * (a >> 1)) if kind else (lambda ib, a: 4 * ib + a)
        assert sorted(out_index(ib, a) for ib in range(4) for a in range(4)) == list(range(16))
        assert sorted(8 * c + in_index(kb, e) for c in range(2) for kb in range(4) for e in range(2)) == list(range(16))
        assert all(out_index(ib, a) % 2 == a % 2 for ib in range(4) for a in range(4))
        for inverse in (0, 1):
            a_img = (C.c_uint8 * 8192)()
            c_img = (C.c_int32 * 128)()
            tw256 = (C.c_uint64 * 4096)()
            tw16 = (C.c_uint64 * 256)()
            assert L.bp_debug_ntt_mx_tables(kind, inverse, a_img, c_img, tw256, tw16) == 0
            a_img = np.frombuffer(a_img, dtype=np.int8).reshape(8, 64, 16)
            c_img = np.frombuffer(c_img, dtype=np.int32).reshape(8, 4, 4)
            w4096 = m.root(12)
            if inverse:
                w4096 = pow(w4096, P - 2, P)
            w16 = pow(w4096, 256, P)
            A, _ = m.dft16_matrix(w16)                     # A[(k, q)][(j, p)]: balanced digits of w16^(jk) 2^(8p)
            base = sum((1 << 22) << (8 * q) for q in range(8))
            delta = (-base) % P
            for rb in range(8):
                a, h = rb >> 1, rb & 1
                for lane in range(64):
                    r, kb = lane & 15, lane >> 4
                    out, q = out_index(r >> 2, a), 4 * h + (r & 3)
                    for b in range(16):
                        assert int(a_img[rb, lane, b]) == A[out * 8 + q][in_index(kb, b >> 3) * 8 + (b & 7)]
                for ib in range(4):
                    for reg in range(4):
                        out, q = out_index(ib, a), 4 * h + reg
                        s0 = sum(A[out * 8 + q][in_index(kb, e) * 8 + p] for kb in range(4) for e in range(2)
                                 for p in range(8))
                        want = (1 << 22) + ((delta >> (8 * q)) & 0xFF) + 128 * s0 + (-127 * s0 if a & 1 else 128 * s0)
                        assert int(c_img[rb, ib, reg]) == want
            # the two-chunk product with these operands IS the 16-point DFT: integer check on random inputs
            rng = np.random.default_rng(9 + kind)
            xs = [int(v) for v in rng.integers(0, 1 << 64, 16, dtype=np.uint64)]
            for ib in range(4):
                for a in range(4):
                    out = out_index(ib, a)
                    planes = []
                    for q in range(8):
                        rb, reg = 2 * a + (q >> 2), q & 3
                        acc = int(c_img[rb, ib, reg])
                        for kb in range(4):
                            for e in range(2):
                                i0 = in_index(kb, e)
                                for p in range(8):
                                    d = A[out * 8 + q][i0 * 8 + p]
                                    x0 = (xs[i0] >> (8 * p)) & 0xFF
                                    x1 = (xs[i0 + 8] >> (8 * p)) & 0xFF
                                    acc += d * (x0 - 128) + d * ((127 - x1) if a & 1 else (x1 - 128))
                        assert 0 <= acc < (1 << 23)
                        planes.append(acc)
                    got = sum(v << (8 * q) for q, v in enumerate(planes)) % P
                    assert got == sum(xs[j] * pow(w16, j * out, P) for j in range(16)) % P
            t256 = np.frombuffer(tw256, dtype=np.uint64).reshape(16, 256)
            t16 = np.frombuffer(tw16, dtype=np.uint64).reshape(16, 16)
            w256 = pow(w4096, 16, P)
            for k in (0, 1, 5, 15):
                for i in (0, 1, 17, 255):
                    assert int(t256[k, i]) == pow(w4096, i * k, P)
                for i in (0, 3, 15):
                    assert int(t16[k, i]) == pow(w256, i * k, P)


# ---- grouped partial rounds of the matrix-core Poseidon kernels (csrc/poseidon_group.cpp) -----------------------
@pytest.mark.parametrize("K,r0", [(8, 4), (8, 12), (6, 20), (4, 8)])
def test_poseidon_group_tables_match_the_integer_model(K, r0):
    """The operand images the library uploads (A operands: balanced base-256 digits of the 64-bit coefficients of the
    grouped rounds' affine forms, in the MFMA lane layout; C operands: bias, non-negativity offsets, round constants)
    are byte for byte those of tools/poseidon_group_model.py -- whose check() runs the kernel's own sequence on these
    images, on exact integers, against the plain permutation.  Host only: no GPU."""
    import ctypes as C
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import poseidon_group_model as model
    import proof_protocol_decoder_amd as pkg
    L = pkg.lib()
    L.bp_debug_poseidon_group_ops.restype = C.c_uint32
    n_ops = L.bp_debug_poseidon_group_ops(K)
    G = model.check(K, r0, n=3)                      # grouped == plain rounds, tile form and device order
    ops, cform, cmain = model.device_images(G)
    assert n_ops == model.layout(K)["n_ops"] == len(ops) // 1024
    got_ops = (C.c_uint8 * (n_ops * 1024))()
    got_cf, got_cm, bound = (C.c_int32 * 64)(), (C.c_int32 * 96)(), C.c_int32()
    assert L.bp_debug_poseidon_group_tables(K, r0, got_ops, got_cf, got_cm, C.byref(bound)) == 0
    assert bytes(got_ops) == ops
    assert list(got_cf) == cform and list(got_cm) == cmain
    assert bound.value == G["bound"] < (1 << 23)
    assert L.bp_debug_poseidon_group_tables(9, 4, got_ops, got_cf, got_cm, None) != 0
    assert L.bp_debug_poseidon_group_tables(8, 20, got_ops, got_cf, got_cm, None) != 0     # would run past round 25


def test_poseidon_group_operands_do_not_depend_on_the_round():
    """Both groups of the kernel (rounds 4..11 and 12..19) share ONE set of A operands; only the C tables differ."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import poseidon_group_model as model
    a, b = model.device_images(model.build_group(8, 4)), model.device_images(model.build_group(8, 12))
    assert a[0] == b[0] and a[1] != b[1] and a[2] != b[2]


def test_poseidon_short_group_runs_on_the_long_groups_form_operands():
    """The third group of the kernels (rounds 20..25): steps 0..5 on the EIGHT-round group's form operands, the new
    state from a six-round group's MAIN operands and C tables -- the kernel's sequence on exact integers against the
    plain rounds (the library checks while it builds the image that the six-round group's form operands are the
    eight-round group's with the rows of forms 6 and 7 blank)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import poseidon_group_model as model
    assert model.check_short(n=4)
    # the same blank-row relation on the model's images
    ops8, ops6 = model.device_images(model.build_group(8, 4))[0], model.device_images(model.build_group(6, 20))[0]
    L6 = model.layout(6)
    for i in range(L6["main_base"]):
        for b in range(1024):
            blank = i >= L6["w_base"][1] and ((b >> 4) & 15) >= 8
            assert ops6[i * 1024 + b] == (0 if blank else ops8[i * 1024 + b]), (i, b)
