#!/usr/bin/env python3
"""Integer model of the GROUPED partial rounds of the matrix-core Poseidon kernels (csrc/poseidon_mx.cuh, round 3).

In a partial round only state word 0 meets an S-box; the other eleven words evolve linearly.  Taking K rounds at a
time (K <= 8), with t(r0) = (t0, w) the S-box-input form of the state at the group's first round r0 and
sigma_j = sbox(t0(r0 + j)):

    F_j = t0(r0 + j)   = <a_j, w> + sum_{i < j} g_{j,i} sigma_i + const_j          (j = 1 .. K-1; F_0 = t0)
    t(r0 + K)          = N_w w + N_s (sigma_0 .. sigma_{K-1}) + const              (all twelve words)

so per round the device recombines ONE word (the form F_j, out of an int8 MFMA over the bytes of w and of the
earlier sigmas) instead of twelve, and the full twelve-word recombination happens once per group.

Everything the device does with these matrices is modelled here on exact integers: the balanced base-256 digits of
the 64-bit coefficients (A operands of v_mfma_i32_16x16x64_i8), the -128 bias of the byte operands and the constants
that undo it (C operands), the non-negativity offsets, the plane recombination and the 4-instruction row reduction.
`build_group(K, r0)` returns the operand images; tests/test_mx_tables.py compares them with what the library builds
(bp_debug_poseidon_group_tables) and this file's `check()` compares the grouped rounds with the plain permutation.

Operand geometry (poseidon_mx.cuh): lane (n, kb) of a wave holds words kb, kb+4, kb+8 of state n.  K index of an MFMA
operand: k = 16*kb + 4*dword + byte.
  chunk LO : dword a (0..2) = low half of word kb+4a, byte = plane 0..3;  dword 3 unused
  chunk HI : the high halves (planes 4..7)
  chunk SIG: dwords (0,1) = sigma_kb (lo, hi), dwords (2,3) = sigma_{kb+4}
Tiles: 16 rows; row r of a result lands in lane group r >> 2, register r & 3.
  form tile pair P (forms 4P .. 4P+3): tile L row 4(f%4)+p = plane p of form f (p < 4), tile H likewise planes 4..7
  main tile (g, h): row 4*ib + p = plane 4h+p of output word ib + 4g
"""
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


def sbox(x):
    return pow(x, 7, P)


# ---- affine forms over the variables (w_1..w_11, sigma_0..sigma_{K-1}, 1) -------------------------------------
def affine_group(K, r0):
    """-> forms[j] (j = 0..K-1; forms[0] is None: F_0 = t0 itself) and out[i] (the 12 words of t(r0 + K)); each an
    affine map as (coef over 11 + K variables, const)."""
    nv = 11 + K
    zero = lambda: [0] * nv
    unit = lambda i: [int(k == i) for k in range(nv)]
    # t(r0): t0 is the S-box input (not a variable), w_i = variable i-1
    w = [(unit(i), 0) for i in range(11)]
    forms = [None]
    for j in range(K):
        sig = (unit(11 + j), 0)
        t = [sig] + w                                   # the state after the S-box of round r0 + j
        nxt = []
        for i in range(12):
            coef, const = zero(), 0
            for k in range(12):
                if M[i][k]:
                    coef = [(a + M[i][k] * b) % P for a, b in zip(coef, t[k][0])]
                    const = (const + M[i][k] * t[k][1]) % P
            if r0 + j + 1 < 30:
                const = (const + RC[r0 + j + 1][i]) % P
            nxt.append((coef, const))
        if j + 1 < K:
            forms.append(nxt[0])
        w = nxt[1:]
        last = nxt
    return forms, last


# ---- digits -----------------------------------------------------------------------------------------------------
def balanced_digits(d):
    """d in [0, p) -> eight digits in [-128, 127] of d or d - p (whichever fits), least significant first"""
    v = d if d <= P // 2 else d - P
    out = []
    for _ in range(8):
        r = ((v + 128) % 256) - 128
        out.append(r)
        v = (v - r) // 256
    assert v == 0, d
    return out


def var_of_k(chunk, k, K):
    """operand byte k of a chunk -> (variable index, byte q) or None.  Variables: 0..10 = w_1..w_11, 11+j = sigma_j"""
    kb, dword, byte = k >> 4, (k >> 2) & 3, k & 3
    if chunk in ("LO", "HI"):
        if dword == 3:
            return None
        word = kb + 4 * dword
        if word == 0:
            return None                                 # word 0 holds t0, which only feeds the S-box
        return word - 1, byte + (4 if chunk == "HI" else 0)
    j = kb + 4 * (dword >> 1)
    if j >= K:
        return None
    return 11 + j, byte + 4 * (dword & 1)


CHUNKS = ("LO", "HI", "SIG")


def tile_rows_forms(P_idx, half):
    """row -> (form, plane) of a form tile"""
    return [(4 * P_idx + (r >> 2), 4 * half + (r & 3)) for r in range(16)]


def tile_rows_main(g, h):
    return [((r >> 2) + 4 * g, 4 * h + (r & 3)) for r in range(16)]


def build_tile(exprs, rows, K):
    """exprs[target] = (coef, const) or None.  -> A[chunk] (16 x 64 digits), C (16 ints), per-row bounds.
    sum_p S[target, p] * 256^p == expr (mod p) with S = C + sum_k A[k] * (byte_k - 128), all S >= 0."""
    A = {c: [[0] * 64 for _ in range(16)] for c in CHUNKS}
    digs = {}
    for r, (target, plane) in enumerate(rows):
        e = exprs[target] if target < len(exprs) else None
        if e is None:
            continue
        for c in CHUNKS:
            for k in range(64):
                vq = var_of_k(c, k, K)
                if vq is None:
                    continue
                v, q = vq
                if e[0][v] == 0:
                    continue
                key = (target, v, q)
                if key not in digs:
                    digs[key] = balanced_digits(e[0][v] * (1 << (8 * q)) % P)
                A[c][r][k] = digs[key][plane]
    return A


def finish_constants(exprs, tiles, K):
    """tiles: list of (rows, A) covering all 8 planes of every target.  Computes C per tile so that every plane sum is
    non-negative: S_p = sum_k A*(x-128) + C_p with C_p = 128*rowsum + (-lo_p) + cdigit_p, where lo_p is the least
    value of sum A*x and the cdigits (0..255) spell const - sum_p (-lo_p) 256^p (mod p)."""
    lo, rowsum = {}, {}
    for rows, A in tiles:
        for r, (target, plane) in enumerate(rows):
            neg = sum(min(a, 0) for c in CHUNKS for a in A[c][r])
            lo[(target, plane)] = 255 * neg
            rowsum[(target, plane)] = sum(a for c in CHUNKS for a in A[c][r])
    out = []
    hi_bound = 0
    for rows, A in tiles:
        C = [0] * 16
        for r, (target, plane) in enumerate(rows):
            e = exprs[target] if target < len(exprs) else None
            if e is None:
                continue
            shift = sum((-lo[(target, p)]) << (8 * p) for p in range(8))
            cd = (e[1] - shift) % P
            cdig = [(cd >> (8 * p)) & 0xFF for p in range(8)]
            C[r] = 128 * rowsum[(target, plane)] - lo[(target, plane)] + cdig[plane]
            pos = sum(max(a, 0) for c in CHUNKS for a in A[c][r])
            hi_bound = max(hi_bound, 255 * pos - lo[(target, plane)] + 255)
        out.append(C)
    return out, hi_bound


def build_group(K, r0):
    forms, last = affine_group(K, r0)
    n_pairs = (K + 3) // 4
    tiles = []           # (name, rows, A)
    for Pi in range(n_pairs):
        for half in range(2):
            rows = tile_rows_forms(Pi, half)
            tiles.append((("form", Pi, half), rows, build_tile(forms, rows, K)))
    form_C, form_bound = finish_constants(forms, [(t[1], t[2]) for t in tiles], K)
    mt = []
    for g in range(3):
        for h in range(2):
            rows = tile_rows_main(g, h)
            mt.append((("main", g, h), rows, build_tile(last, rows, K)))
    main_C, main_bound = finish_constants(last, [(t[1], t[2]) for t in mt], K)
    return {"K": K, "r0": r0, "forms": forms, "last": last, "form_tiles": tiles, "form_C": form_C,
            "main_tiles": mt, "main_C": main_C, "bound": max(form_bound, main_bound)}


# ---- the device arithmetic on integers -------------------------------------------------------------------------
def mfma(A, Bbytes, C):
    """D[row] = C[row] + sum_k A[row][k] * (B[k] - 128) for one column"""
    return [C[r] + sum(a * (b - 128) for a, b in zip(A[r], Bbytes) if a) for r in range(16)]


def chunk_bytes(chunk, t, sig, K):
    """operand bytes (64) of one state: t = 12 words (t0 in slot 0), sig = list of known sigmas (None = garbage)"""
    out = []
    for k in range(64):
        kb, dword, byte = k >> 4, (k >> 2) & 3, k & 3
        if chunk in ("LO", "HI"):
            if dword == 3:
                out.append(0x5A)                       # unused dword: anything
                continue
            word = t[kb + 4 * dword]
            out.append((word >> (8 * (byte + (4 if chunk == "HI" else 0)))) & 0xFF)
        else:
            j = kb + 4 * (dword >> 1)
            s = sig[j] if j < len(sig) and sig[j] is not None else 0xA5A5A5A5A5A5A5A5   # not yet known: garbage
            out.append((s >> (8 * (byte + 4 * (dword & 1)))) & 0xFF)
    return out


def recombine(planes_lo, planes_hi, bound):
    """planes64 + reduce_rows on integers, with the bounds the device code relies on"""
    for s in planes_lo + planes_hi:
        assert 0 <= s < (1 << 23) and s <= bound      # mxa::planes wants plane sums below 2^23
    L = sum(s << (8 * i) for i, s in enumerate(planes_lo))
    H = sum(s << (8 * i) for i, s in enumerate(planes_hi))
    assert L < (1 << 49) and H < (1 << 49)
    # reduce_rows: H = h0 + h1*2^32; X = L + h1*EPS (no carry); hi word + h0 may carry once (weight 2^64 = EPS)
    h0, h1 = H & 0xFFFFFFFF, H >> 32
    X = L + h1 * 0xFFFFFFFF
    assert X < (1 << 64)
    lo, hi = X & 0xFFFFFFFF, (X >> 32) + h0
    c = hi >> 32
    hi &= 0xFFFFFFFF
    v = lo + (hi << 32) + c * 0xFFFFFFFF
    assert v < (1 << 64)                                # the device result is a lazily reduced u64
    return v % P


def run_group(G, t):
    """t: the twelve words of t(r0) (any u64 < p).  -> (t(r0 + K) via the operand images, sigmas)"""
    K = G["K"]
    sig = [None] * K
    lo, hi = chunk_bytes("LO", t, sig, K), chunk_bytes("HI", t, sig, K)
    acc = {}
    for (name, rows, A), C in zip(G["form_tiles"], G["form_C"]):
        d = mfma(A["LO"], lo, C)
        d = mfma(A["HI"], hi, d)
        acc[name] = d
    t0 = t[0]
    for j in range(K):
        if j:
            Pi, f = j // 4, j % 4
            t0 = recombine(acc[("form", Pi, 0)][4 * f:4 * f + 4], acc[("form", Pi, 1)][4 * f:4 * f + 4], G["bound"])
        sig[j] = sbox(t0)
        sb = chunk_bytes("SIG", t, sig, K)
        # delta: sigma_j into every form tile that still holds a form f > j.  The A image of chunk SIG restricted to
        # sigma_j's bytes is what the device loads as A_delta(P, j).
        for (name, rows, A), _ in zip(G["form_tiles"], G["form_C"]):
            if 4 * name[1] + 3 <= j:
                continue
            Aj = [[a if var_of_k("SIG", k, K) and var_of_k("SIG", k, K)[0] == 11 + j else 0 for k, a in enumerate(row)]
                  for row in A["SIG"]]
            acc[name] = mfma(Aj, sb, acc[name])
    sb = chunk_bytes("SIG", t, sig, K)
    out = [0] * 12
    res = {}
    for (name, rows, A), C in zip(G["main_tiles"], G["main_C"]):
        d = mfma(A["LO"], lo, C)
        d = mfma(A["HI"], hi, d)
        d = mfma(A["SIG"], sb, d)
        res[name] = d
    for g in range(3):
        for ib in range(4):
            out[ib + 4 * g] = recombine(res[("main", g, 0)][4 * ib:4 * ib + 4], res[("main", g, 1)][4 * ib:4 * ib + 4],
                                        G["bound"])
    return out, sig


# ---- the operand images in the device's layout and order (csrc/poseidon_group.hpp) ------------------------------
def layout(K):
    n_pairs = (K + 3) // 4
    L = {"K": K, "n_pairs": n_pairs, "w_base": [], "w_per_half": [], "d_base": [], "d_first": [], "d_count": []}
    idx = 0
    for Pi in range(n_pairs):
        L["w_base"].append(idx)
        L["w_per_half"].append(3 if Pi else 2)
        idx += 2 * L["w_per_half"][Pi]
        last_form = min(4 * Pi + 3, K - 1)
        L["d_first"].append(4 * Pi)
        L["d_count"].append(max(last_form - 4 * Pi, 0))
        L["d_base"].append(idx)
        idx += 2 * L["d_count"][Pi]
    L["main_base"] = idx
    L["n_ops"] = idx + 18
    return L


def operand_image(A, K, j_only=None, j_below=None):
    """16 x 64 digits -> 1024 bytes: lane l = (row = l & 15, kblock = l >> 4) holds A[row][16 kblock .. + 15]"""
    out = bytearray()
    for l in range(64):
        for b in range(16):
            k = 16 * (l >> 4) + b
            a = A[l & 15][k]
            if j_only is not None or j_below is not None:
                vq = var_of_k("SIG", k, K)
                j = vq[0] - 11 if vq else -1
                if vq is None or (j_only is not None and j != j_only) or (j_below is not None and j >= j_below):
                    a = 0
            out.append(a & 0xFF)
    return bytes(out)


def device_images(G):
    """-> (ops bytes, cform (64 ints), cmain (96 ints)) exactly as poseidon::group::build lays them out"""
    K, L = G["K"], layout(G["K"])
    ft = {name: A for name, rows, A in G["form_tiles"]}
    ops = bytearray()
    for Pi in range(L["n_pairs"]):
        for half in range(2):
            A = ft[("form", Pi, half)]
            ops += operand_image(A["LO"], K) + operand_image(A["HI"], K)
            if Pi:
                ops += operand_image(A["SIG"], K, j_below=4 * Pi)
        for d in range(L["d_count"][Pi]):
            for half in range(2):
                ops += operand_image(ft[("form", Pi, half)]["SIG"], K, j_only=L["d_first"][Pi] + d)
    for (name, rows, A) in G["main_tiles"]:
        for c in CHUNKS:
            ops += operand_image(A[c], K)
    assert len(ops) == L["n_ops"] * 1024
    cform = [0] * 64
    for (name, rows, A), C in zip(G["form_tiles"], G["form_C"]):
        for r in range(16):
            cform[(name[1] * 2 + name[2]) * 16 + r] = C[r]
    cmain = [0] * 96
    for i, C in enumerate(G["main_C"]):
        for r in range(16):
            cmain[i * 16 + r] = C[r]
    return bytes(ops), cform, cmain


def image_rows(ops, idx):
    """operand idx of an image back as 16 x 64 signed digits"""
    A = [[0] * 64 for _ in range(16)]
    for l in range(64):
        for b in range(16):
            v = ops[idx * 1024 + l * 16 + b]
            A[l & 15][16 * (l >> 4) + b] = v - 256 if v >= 128 else v
    return A


def run_group_device(K, ops, cform, cmain, t, bound):
    """The kernel's own sequence (poseidon_mx.cuh, grp::partial_group) on the operand IMAGES: a pair of form tiles is
    started at step 4P from the W operands (and the sigmas known by then), every later sigma is added by a D operand,
    the new state comes from the 18 MAIN operands."""
    L = layout(K)
    sig = [None] * K
    lo, hi = chunk_bytes("LO", t, sig, K), chunk_bytes("HI", t, sig, K)
    acc = [None, None]
    t0 = t[0]
    for j in range(K):
        Pi, f = j // 4, j % 4
        if f == 0:
            nw = L["w_per_half"][Pi]
            sb = chunk_bytes("SIG", t, sig, K)
            for half in range(2):
                base = L["w_base"][Pi] + half * nw
                d = mfma(image_rows(ops, base), lo, cform[(Pi * 2 + half) * 16:(Pi * 2 + half) * 16 + 16])
                d = mfma(image_rows(ops, base + 1), hi, d)
                if Pi:
                    d = mfma(image_rows(ops, base + 2), sb, d)
                acc[half] = d
        if j:
            t0 = recombine(acc[0][4 * f:4 * f + 4], acc[1][4 * f:4 * f + 4], bound)
        sig[j] = sbox(t0)
        if L["d_first"][Pi] <= j < L["d_first"][Pi] + L["d_count"][Pi]:
            sb = chunk_bytes("SIG", t, sig, K)
            for half in range(2):
                acc[half] = mfma(image_rows(ops, L["d_base"][Pi] + 2 * (j - L["d_first"][Pi]) + half), sb, acc[half])
    sb = chunk_bytes("SIG", t, sig, K)
    out = [0] * 12
    for g in range(3):
        res = []
        for h in range(2):
            base = L["main_base"] + (g * 2 + h) * 3
            d = mfma(image_rows(ops, base), lo, cmain[(g * 2 + h) * 16:(g * 2 + h) * 16 + 16])
            d = mfma(image_rows(ops, base + 1), hi, d)
            d = mfma(image_rows(ops, base + 2), sb, d)
            res.append(d)
        for ib in range(4):
            out[ib + 4 * g] = recombine(res[0][4 * ib:4 * ib + 4], res[1][4 * ib:4 * ib + 4], bound)
    return out, sig


def run_short_group_device(ops8, main6, cform6, cmain6, t, bound, K=6):
    """The kernel's THIRD group (rounds 20..25, poseidon_mx.cuh: partial_group with full = false): the form operands
    are the eight-round group's (rows of forms 6 and 7 come out as garbage nobody reads; the delta of sigma_5, which
    only they need, is skipped), the new state comes from the MAIN operands of a six-round group, the C tables are
    the six-round group's."""
    L = layout(8)
    sig = [None] * 8
    lo, hi = chunk_bytes("LO", t, sig, 8), chunk_bytes("HI", t, sig, 8)
    acc = [None, None]
    t0 = t[0]
    for j in range(K):
        Pi, f = j // 4, j % 4
        if f == 0:
            nw = L["w_per_half"][Pi]
            sb = chunk_bytes("SIG", t, sig, 8)
            for half in range(2):
                base = L["w_base"][Pi] + half * nw
                d = mfma(image_rows(ops8, base), lo, cform6[(Pi * 2 + half) * 16:(Pi * 2 + half) * 16 + 16])
                d = mfma(image_rows(ops8, base + 1), hi, d)
                if Pi:
                    d = mfma(image_rows(ops8, base + 2), sb, d)
                acc[half] = d
        if j:
            t0 = recombine(acc[0][4 * f:4 * f + 4], acc[1][4 * f:4 * f + 4], bound)
        sig[j] = sbox(t0)
        if L["d_first"][Pi] <= j < L["d_first"][Pi] + L["d_count"][Pi] and j != K - 1:
            sb = chunk_bytes("SIG", t, sig, 8)
            for half in range(2):
                acc[half] = mfma(image_rows(ops8, L["d_base"][Pi] + 2 * (j - L["d_first"][Pi]) + half), sb, acc[half])
    sb = chunk_bytes("SIG", t, sig, 8)
    out = [0] * 12
    for g in range(3):
        res = []
        for h in range(2):
            base = (g * 2 + h) * 3
            d = mfma(image_rows(main6, base), lo, cmain6[(g * 2 + h) * 16:(g * 2 + h) * 16 + 16])
            d = mfma(image_rows(main6, base + 1), hi, d)
            d = mfma(image_rows(main6, base + 2), sb, d)
            res.append(d)
        for ib in range(4):
            out[ib + 4 * g] = recombine(res[0][4 * ib:4 * ib + 4], res[1][4 * ib:4 * ib + 4], bound)
    return out, sig[:K]


def short_group_images(r0=20):
    """-> (ops8, main6, cform6, cmain6, bound): what the three-group kernels hold for their last group"""
    G8, G6 = build_group(8, 4), build_group(6, r0)
    ops8 = device_images(G8)[0]
    ops6, cform6, cmain6 = device_images(G6)
    mb = layout(6)["main_base"]
    return ops8, ops6[mb * 1024:(mb + 18) * 1024], cform6, cmain6, max(G8["bound"], G6["bound"])


def check_short(n=6, seed=2, r0=20):
    ops8, main6, cform6, cmain6, bound = short_group_images(r0)
    rng = random.Random(seed)
    cases = [[rng.randrange(P) for _ in range(12)] for _ in range(n)]
    cases += [[0] * 12, [P - 1] * 12, [0xFFFFFFFF00000000] * 12, [0x00000000FFFFFFFF] * 12]
    for t in cases:
        want, ws = plain_rounds(list(t), r0, 6)
        got, gs = run_short_group_device(ops8, main6, cform6, cmain6, list(t), bound)
        assert gs == ws and got == want, ("short group", r0)
    return True


def plain_rounds(t, r0, K):
    sig = []
    for r in range(r0, r0 + K):
        s = sbox(t[0])
        sig.append(s)
        u = [s] + t[1:]
        t = [(sum(M[i][k] * u[k] for k in range(12)) + (RC[r + 1][i] if r + 1 < 30 else 0)) % P for i in range(12)]
    return t, sig


def check(K, r0, n=6, seed=1):
    G = build_group(K, r0)
    rng = random.Random(seed)
    cases = [[rng.randrange(P) for _ in range(12)] for _ in range(n)]
    cases += [[0] * 12, [P - 1] * 12, [0xFFFFFFFF00000000] * 12, [0x00000000FFFFFFFF] * 12]
    ops, cform, cmain = device_images(G)
    for t in cases:
        want, ws = plain_rounds(list(t), r0, K)
        got, gs = run_group(G, list(t))
        assert gs == ws and got == want, (K, r0)
        got, gs = run_group_device(K, ops, cform, cmain, list(t), G["bound"])
        assert gs == ws and got == want, ("device order", K, r0)
    return G


if __name__ == "__main__":
    for K, r0 in ((8, 4), (8, 12), (6, 20), (4, 4)):
        G = check(K, r0)
        print("K=%d r0=%d: grouped rounds == plain rounds (tile form and the device's operand images, %d operands); "
              "largest plane sum %d (< 2^23 = %d)" % (K, r0, layout(K)["n_ops"], G["bound"], 1 << 23))
    check_short()
    print("rounds 20..25 on the eight-round group's form operands + a six-round group's MAIN operands == plain rounds")
