#!/usr/bin/env python3
"""Integer model of the matrix-core NTT passes (csrc/ntt_mx.cuh): checks, with Python integers, that

  * a 16-point DFT over Goldilocks is an exact int8 matrix product on the BYTES of its inputs: entry
    [(k, q)][(j, p)] = balanced base-256 digit q of w16^(j k) * 2^(8p) mod p (every such constant is +-2^e or
    +-(2^a - 2^b) because 2 is a 192nd root of unity and w16 = 2^156), bytes taken as signed x - 128, row constants
    (a bias that is a multiple of p and makes every plane sum non-negative, plus 128 * the row's digit sum) in the
    accumulator, plane sums recombined with shifts;
  * three such passes with the twiddle tables TW0 / TW1 and the in-place bit-reversed placement of the DFT outputs
    make the 4096-point decimation-in-frequency transform (natural in, bit-reversed out), and the transposed
    pipeline the decimation-in-time one (bit-reversed in, natural out);
  * the bounds the device code relies on (plane sums < 2^23, recombined halves < 2^48).

Run: python tools/ntt_mx_model.py   (a few seconds; prints 'ok')."""
import random

P = 0xFFFFFFFF00000001
ROOT32 = 1753635133440165772  # 7^((p-1)/2^32)


def root(k):
    r = ROOT32
    for _ in range(k, 32):
        r = r * r % P
    return r


def digits8(w):
    """balanced base-256 digits d[0..7] in [-128, 127] of w or w - P (whichever fits eight digits)"""
    lim = 127 * ((1 << 64) - 1) // 255
    v = w if w <= lim else w - P
    d = []
    for _ in range(8):
        x = ((v + 128) % 256) - 128
        d.append(x)
        v = (v - x) >> 8
    assert v == 0, "does not fit eight balanced digits"
    return d


def dft16_matrix(w16):
    """A[(k, q)][(j, p)], row constants C[(k, q)] (bias + offset correction), as nested lists"""
    A = [[0] * 128 for _ in range(128)]
    for k in range(16):
        for j in range(16):
            for p in range(8):
                d = digits8(pow(w16, j * k, P) * (1 << (8 * p)) % P)
                for q in range(8):
                    A[k * 8 + q][j * 8 + p] = d[q]
    # bias: B_q = 2^22 + delta_q with sum B_q 2^(8q) = 0 (mod p), so that every plane sum is >= 0
    base = sum((1 << 22) << (8 * q) for q in range(8))
    delta = (-base) % P
    # delta as eight non-negative digits < 2^8 ... it may need a ninth; spread it: digits of delta in base 256 (delta < 2^64)
    dd = [(delta >> (8 * q)) & 0xFF for q in range(8)]
    C = []
    for k in range(16):
        for q in range(8):
            C.append((1 << 22) + dd[q] + 128 * sum(A[k * 8 + q]))
    return A, C


def mfma_dft16(A, C, xs):
    """xs: 16 field elements (any u64).  Returns the 16 outputs, computed the way the device does."""
    b = []
    for x in xs:
        for p in range(8):
            b.append(((x >> (8 * p)) & 0xFF) - 128)   # the XOR 0x80 turns the byte into this signed value
    out = []
    for k in range(16):
        s = [C[k * 8 + q] + sum(A[k * 8 + q][c] * b[c] for c in range(128)) for q in range(8)]
        assert all(0 <= v < (1 << 23) for v in s), s
        e, f = s[0] + (s[1] << 8), s[2] + (s[3] << 8)
        g, h = s[4] + (s[5] << 8), s[6] + (s[7] << 8)
        assert max(e, f, g, h) < (1 << 32)
        L, H = e + (f << 16), g + (h << 16)
        assert L < (1 << 48) and H < (1 << 48)
        out.append((L + (H << 32)) % P)
    return out


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def check_dft16(w16):
    A, C = dft16_matrix(w16)
    rng = random.Random(1)
    for trial in range(6):
        xs = [rng.randrange(1 << 64) for _ in range(16)]
        if trial == 0:
            xs = [0] * 16
        if trial == 1:
            xs = [(1 << 64) - 1] * 16
        if trial == 2:
            xs = [0x8080808080808080] * 16
        got = mfma_dft16(A, C, xs)
        want = [sum(xs[j] * pow(w16, j * k, P) for j in range(16)) % P for k in range(16)]
        assert got == want
    return A, C


def dif4096(x, w, A, C):
    """natural in, bit-reversed out; w = primitive 4096th root.  Three passes of S = 4096, 256, 16."""
    n = 4096
    buf = list(x)
    for S in (4096, 256, 16):
        s = S // 16
        ws = pow(w, n // S, P)
        nxt = [0] * n
        for blk in range(n // S):
            for i in range(s):
                xs = [buf[blk * S + j * s + i] for j in range(16)]
                ys = mfma_dft16(A, C, xs)
                for k in range(16):
                    nxt[blk * S + bitrev(k, 4) * s + i] = ys[k] * pow(ws, i * k, P) % P   # TW[S][k][i]; 1 for S = 16
        buf = nxt
    return buf


def dit4096(c_br, w, A, C):
    """bit-reversed in, natural out: the transposed pipeline, S = 16, 256, 4096."""
    n = 4096
    buf = list(c_br)
    for S in (16, 256, 4096):
        s = S // 16
        ws = pow(w, n // S, P)
        nxt = [0] * n
        for blk in range(n // S):
            for i in range(s):
                xs = [buf[blk * S + bitrev(k, 4) * s + i] * pow(ws, i * k, P) % P for k in range(16)]
                ys = mfma_dft16(A, C, xs)
                for j in range(16):
                    nxt[blk * S + j * s + i] = ys[j]
        buf = nxt
    return buf


def main():
    assert root(6) == pow(2, 39, P) and root(4) == pow(2, 156, P)   # w64 = 2^39, w16 = 2^156 = -2^60: powers of two
    for w16 in (root(4), pow(root(4), P - 2, P)):
        A, C = check_dft16(w16)
        assert all(-128 <= v <= 127 for row in A for v in row)
    # whole transforms on a reduced size (the pipeline is size-generic in S): 4096 points, forward and inverse roots
    rng = random.Random(2)
    for inverse in (False, True):
        w = root(12)
        if inverse:
            w = pow(w, P - 2, P)
        A, C = dft16_matrix(pow(w, 256, P))
        x = [rng.randrange(P) for _ in range(4096)]
        # naive DFT at a handful of output indices (full O(n^2) is 16M big-int products: sample 24)
        got = dif4096(x, w, A, C)
        for k in [0, 1, 2, 3, 255, 256, 1000, 2048, 4095] + [rng.randrange(4096) for _ in range(15)]:
            want = sum(x[i] * pow(w, i * k, P) for i in range(4096)) % P
            assert got[bitrev(k, 12)] == want, k
        # DIT on the bit-reversed coefficients gives the same evaluations in natural order
        c_br = [x[bitrev(i, 12)] for i in range(4096)]
        nat = dit4096(c_br, w, A, C)
        assert [nat[bitrev(i, 12)] for i in range(4096)] == got
    print("ok")




# ---- LDS image: bank-conflict check of the XOR swizzle the kernels use (csrc/ntt_mx.cuh, swz12) ------------------
def swz12(pos):
    return pos ^ ((pos >> 4) & 15) ^ ((((pos >> 1) ^ (pos >> 3) ^ (pos >> 9) ^ (pos >> 11)) & 1) << 4)


def check_swizzle():
    """Every 8-byte LDS access of a pass, per half wave (32 lanes, one 256-byte row of the 64 banks): all 32 lanes
    must fall on different 8-byte bank pairs.  Lane = (n = lane & 15, kb or ib = lane >> 4)."""
    assert sorted(swz12(p) for p in range(4096)) == list(range(4096))   # a permutation
    br4 = lambda k: bitrev(k, 4)
    pats = []
    for w in range(4):
        for m in range(4):
            G = 64 * w + 16 * m
            for e in range(4):       # reads: element j = 8c + 2kb + eps; writes: output k = ib + 4a
                c, eps = e >> 1, e & 1
                j = lambda kb: 8 * c + 2 * kb + eps
                # DIF: slot (c, kb, eps) = input j = 8c + 2kb + eps read at field j, row (ib, a) = output k = 4 ib + a written
                # at field bitrev4(k).  DIT: slot = input kin = 8c + 4 eps + kb read at field bitrev4(kin), row = output
                # j = (a & 1) + 2 ib + 8 (a >> 1) written at field j.
                kd = lambda ib: 4 * ib + e
                kin = lambda kb: 8 * c + 4 * eps + kb
                jt = lambda ib: (e & 1) + 2 * ib + 8 * (e >> 1)
                blk = 4 * w + m
                for tag, rf, wf in (("dif", j, lambda q: br4(kd(q))), ("dit", lambda q: br4(kin(q)), jt)):
                    pats.append(("A rd " + tag, lambda n, q, G=G, rf=rf: rf(q) * 256 + G + n))
                    pats.append(("A wr " + tag, lambda n, q, G=G, wf=wf: wf(q) * 256 + G + n))
                    pats.append(("B rd " + tag, lambda n, q, blk=blk, rf=rf: blk * 256 + rf(q) * 16 + n))
                    pats.append(("B wr " + tag, lambda n, q, blk=blk, wf=wf: blk * 256 + wf(q) * 16 + n))
                    pats.append(("C rd " + tag, lambda n, q, G=G, rf=rf: (G + n) * 16 + rf(q)))
                    pats.append(("C wr " + tag, lambda n, q, G=G, wf=wf: (G + n) * 16 + wf(q)))
    for i in range(16):              # coalesced staging: lane t of 256 takes position i * 256 + t
        for w in range(4):
            pats.append(("stage", lambda n, q, i=i, w=w: i * 256 + 64 * w + 16 * q + n))
    worst = {}
    for name, f in pats:
        for half in range(2):
            banks = [swz12(f(n, q)) & 31 for q in (2 * half, 2 * half + 1) for n in range(16)]
            worst[name] = max(worst.get(name, 0), 32 - len(set(banks)))
    assert all(v == 0 for v in worst.values()), worst
    print("swizzle ok:", ", ".join(sorted(worst)))


if __name__ == "__main__":
    main()
    check_swizzle()
