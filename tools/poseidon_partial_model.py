#!/usr/bin/env python3
"""Integer model of a cheaper form of Poseidon's 22 partial rounds for the matrix-core kernels (DESIGN.md section 7,
"what is left"): groundwork for a later round, checked here against the plain permutation.

In a partial round only word 0 meets an S-box; words 1..11 ("w") evolve linearly.  With t = state + round constant,
sigma_r = sbox(t_0) in round r, and M = [[m00, m0w], [mw0, Mww]]:

    t_0(r+1) = <a_r, w(0)> + sum_{t <= r} g_{r-1-t} * sigma_t + const_r,     a_r = m0w Mww^r,  g_d = m0w Mww^d mw0,  g_{-1} = m00
    w(22)    = Mww^22 w(0) + sum_t (Mww^(21-t) mw0) * sigma_t + const

so the S-box inputs of all 22 rounds are linear forms of the 11 words at the start of the partial rounds and of the
earlier S-box outputs, with a TOEPLITZ dependence on the latter (coefficient depends on r - t only: one constant
operand serves every round if the sigma history is kept most-recent-first).  Per round the device would then
recombine ONE word instead of twelve; the 11 other words are reconstructed once at the end.

This script (a) checks the formulation against the plain rounds on random states, (b) prints the sizes of the
byte-digit matrices an int8-MFMA implementation would need.  Run: python tools/poseidon_partial_model.py"""
import os
import random
import re

P = 0xFFFFFFFF00000001
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MDS_C = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
M = [[(MDS_C[(k - i) % 12] + (8 if i == 0 and k == 0 else 0)) for k in range(12)] for i in range(12)]


def round_constants():
    txt = open(os.path.join(ROOT, "proof_protocol_decoder_amd", "csrc", "poseidon_rc.inc")).read()
    rc = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{16})ULL", txt)]
    assert len(rc) == 360
    return [rc[12 * r:12 * r + 12] for r in range(30)]


RC = round_constants()


def mat_vec(A, v):
    return [sum(a * x for a, x in zip(row, v)) % P for row in A]


def mat_mul(A, B):
    return [[sum(A[i][k] * B[k][j] for k in range(len(B))) % P for j in range(len(B[0]))] for i in range(len(A))]


def sbox(x):
    return pow(x, 7, P)


def partial_rounds_plain(s):
    """rounds 4..25 of the permutation: s is the state entering round 4 (before its constants)"""
    sigmas = []
    for r in range(4, 26):
        t = [(x + c) % P for x, c in zip(s, RC[r])]
        sigmas.append(sbox(t[0]))
        t[0] = sigmas[-1]
        s = mat_vec(M, t)
    return s, sigmas


def build_forms():
    """a_r, g_d, the final maps and all constants, for t(4) = s + RC[4] as the starting point"""
    m00 = M[0][0]
    m0w = [M[0][1:]]                                  # 1 x 11
    mw0 = [[M[i][0]] for i in range(1, 12)]           # 11 x 1
    Mww = [row[1:] for row in M[1:]]                  # 11 x 11
    ident = [[int(i == j) for j in range(11)] for i in range(11)]
    pw = [ident]
    for _ in range(22):
        pw.append(mat_mul(pw[-1], Mww))               # pw[d] = Mww^d
    a = [mat_mul(m0w, pw[r])[0] for r in range(22)]   # a_r (11 entries)
    g = {-1: m00}
    for d in range(21):
        g[d] = mat_mul(mat_mul(m0w, pw[d]), mw0)[0][0]
    return m00, m0w, mw0, Mww, pw, a, g


def partial_rounds_forms(s):
    """the same 22 rounds through the linear forms: returns the state after round 25 and the sigmas"""
    m00, m0w, mw0, Mww, pw, a, g = build_forms()
    t = [(x + c) % P for x, c in zip(s, RC[4])]
    t0, w0 = t[0], t[1:]
    # constants: w(r+1) = Mww w(r) + mw0 sigma_r + rcw(r+1) with rcw(r) = RC[4 + r][1:], t_0(r+1) likewise with RC[..][0]
    # accumulate the constant part of w(r) separately: cw(0) = 0, cw(r+1) = Mww cw(r) + rcw(r+1)
    cw = [[0] * 11]
    for r in range(22):
        nxt = mat_vec(Mww, cw[-1])
        rcn = RC[5 + r][1:] if 5 + r < 30 else [0] * 11
        cw.append([(x + c) % P for x, c in zip(nxt, rcn)])
    sigmas = []
    cur_t0 = t0
    for r in range(22):
        sigmas.append(sbox(cur_t0))
        if r == 21:
            break
        # t_0(r+1) = <a_r, w(0)> + sum_{t<=r} g_{r-1-t} sigma_t + <m0w, cw(r)> + RC[4+r+1][0]
        val = sum(x * y for x, y in zip(a[r], w0))
        val += sum(g[r - 1 - tt] * sigmas[tt] for tt in range(r + 1))
        val += sum(x * y for x, y in zip(m0w[0], cw[r])) + RC[5 + r][0]
        cur_t0 = val % P
    # state after round 25: s = M * t(25) with t(25) = (sigma_21, w(21) incl. its constants)
    w21 = mat_vec(pw[21], w0)
    for tt in range(21):
        col = mat_vec(pw[20 - tt], [row[0] for row in mw0])
        w21 = [(x + c * sigmas[tt]) % P for x, c in zip(w21, col)]
    w21 = [(x + c) % P for x, c in zip(w21, cw[21])]
    return mat_vec(M, [sigmas[21]] + w21), sigmas


def main():
    rng = random.Random(4)
    for _ in range(20):
        s = [rng.randrange(P) for _ in range(12)]
        want, sg_w = partial_rounds_plain(s)
        got, sg_g = partial_rounds_forms(s)
        assert sg_w == sg_g and want == got
    print("22 partial rounds through the linear forms == plain rounds (20 random states)")
    # cost sketch for the int8-MFMA form (per set of 16 states, v_mfma_i32_16x16x64_i8 = 16 rows x K 64 bytes):
    #   forms <a_r, w(0)>: two rounds per tile (16 rows = 2 forms x 8 digits), K = 88 bytes -> 2 K-chunks: 22 MFMAs, operands r-dependent
    #   Toeplitz part: K = 8 bytes per sigma, most recent first: operand constant over r: 3 K-chunks x 4 VGPRs
    #   reconstruction: 11 words x 8 digits = 88 rows (6 tiles), K = 88 + 176 bytes -> 5 chunks: 30 MFMAs
    now = 22 * 6
    new = 22 + 11 * 2 + 30
    print("MFMAs per set for the partial rounds: now %d, with the forms ~%d; words recombined: now %d, then %d"
          % (now, new, 22 * 12, 22 + 11))
    print("operand bytes to stream (not register-resident): forms %d KiB + reconstruction %d KiB per workgroup"
          % (22 * 1024 // 1024, 30 * 1024 // 1024))


if __name__ == "__main__":
    main()
