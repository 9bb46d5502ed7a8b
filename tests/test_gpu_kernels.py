"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from util import (P, bitrev_perm, coset_major_to_natural, rand_field, to_dev, to_host)

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["mx", "valu"])
def ntt_form(request, bpg):
    """2^12..2^14-point blocks: 16-point DFTs on the matrix cores (ntt_mx.cuh) / VALU butterflies"""
    bpg.lib().bp_tune_ntt_mx(2 if request.param == "mx" else 0)   # 2: also the 2^14-point blocks
    yield request.param
    bpg.lib().bp_tune_ntt_mx(3)


@pytest.mark.parametrize("log_n,n_cols", [(0, 3), (1, 2), (2, 5), (3, 4), (5, 7), (9, 16), (12, 9), (13, 3), (14, 5),
                                          (15, 3), (16, 2), (17, 2), (18, 2), (19, 1), (20, 2), (21, 1), (22, 1),
                                          (23, 1)])
def test_ntt_matches_oracle(bpg, oracle, log_n, n_cols, ntt_form):
    rng = np.random.default_rng(100 + log_n)
    n = 1 << log_n
    vals = rand_field(rng, (n_cols, n))
    br = bitrev_perm(log_n)
    # inverse: natural values -> bit-reversed coefficients
    want_coeffs = oracle.ntt_batch(vals, inverse=True)
    got = to_host(bpg.ops.ntt_batch_(to_dev(vals), bpg.ops.NTT_INV_NAT2BR))
    assert (got[:, br] == want_coeffs).all()
    # forward: bit-reversed coefficients -> natural values
    want_vals = oracle.ntt_batch(vals, inverse=False)
    got = to_host(bpg.ops.ntt_batch_(to_dev(vals[:, br]), bpg.ops.NTT_FWD_BR2NAT))
    assert (got == want_vals).all()
    # out of place (two workgroups per 2^13 / 2^14-point block): the same coefficients, input untouched
    d_vals = to_dev(vals)
    got = to_host(bpg.ops.intt_batch(d_vals))
    assert (got[:, br] == want_coeffs).all() and (to_host(d_vals) == vals).all()
    # natural-order API (plonky2 fft/ifft semantics)
    got = to_host(bpg.ops.ntt_batch_(to_dev(vals), bpg.ops.NTT_FWD_NAT))
    assert (got == want_vals).all()
    got = to_host(bpg.ops.ntt_batch_(to_dev(vals), bpg.ops.NTT_INV_NAT))
    assert (got == want_coeffs).all()


@pytest.mark.parametrize("log_n,n_cols", [(9, 4), (12, 5), (13, 3), (14, 2), (15, 2), (18, 1)])
def test_ntt_takes_any_u64_in_every_kernel_form(bpg, log_n, n_cols):
    """include/bpg.h: NTT / LDE inputs may be ANY u64 (reduced mod p on the way in), whichever kernel a launch size
    or tuning knob selects.  The split kernels (two workgroups per 2^13 / 2^14-point block, the default for small
    out-of-place launches) used to feed caller words straight into add_n / sub_n, which need b < p: a = b = 2^64 - 1
    gave 2^32 - 3 instead of 2^33 - 4.  Every form must agree with the transform of the reduced input."""
    rng = np.random.default_rng(900 + log_n)
    n = 1 << log_n
    raw = rng.integers(0, 1 << 64, size=(n_cols, n), dtype=np.uint64)
    raw[:, ::3] = np.uint64(2**64 - 1)            # all-ones and other words >= p, in both halves of every block
    raw[:, 1::5] = np.uint64(P)
    raw[:, n // 2 + 1::7] = np.uint64(P + 12345)
    red = raw % np.uint64(P)
    br = bitrev_perm(log_n)
    try:
        outs = {}
        for mode in (1, 2, 0):                    # never split / split wherever possible / automatic
            bpg.lib().bp_tune_ntt_split(mode)
            for mx in (0, 3):
                bpg.lib().bp_tune_ntt_mx(mx)
                d = to_dev(raw)
                inv = to_host(bpg.ops.intt_batch(d))                      # out of place: split DIF eligible
                assert (to_host(d) == raw).all()
                inv_red = to_host(bpg.ops.intt_batch(to_dev(red)))
                assert (inv == inv_red).all(), ("inverse", mode, mx)
                _, lde = bpg.ops.lde_batch(to_dev(raw[:, br]), 1, from_coeffs=True)   # unscaled + scaled split DIT
                _, lde_red = bpg.ops.lde_batch(to_dev(red[:, br]), 1, from_coeffs=True)
                assert (to_host(lde) == to_host(lde_red)).all(), ("lde", mode, mx)
                fwd = to_host(bpg.ops.ntt_batch_(to_dev(raw[:, br]), bpg.ops.NTT_FWD_BR2NAT))
                fwd_red = to_host(bpg.ops.ntt_batch_(to_dev(red[:, br]), bpg.ops.NTT_FWD_BR2NAT))
                assert (fwd == fwd_red).all(), ("forward", mode, mx)
                outs[(mode, mx)] = (inv, to_host(lde), fwd)
                assert (inv < np.uint64(P)).all() and (fwd < np.uint64(P)).all()
        first = outs[(1, 0)]
        for k, v in outs.items():
            assert all((a == b).all() for a, b in zip(first, v)), k
    finally:
        bpg.lib().bp_tune_ntt_split(0)
        bpg.lib().bp_tune_ntt_mx(3)


@pytest.mark.parametrize("log_n,rate_bits,n_cols", [(3, 1, 2), (6, 1, 5), (9, 1, 16), (12, 3, 7), (14, 1, 4),
                                                    (13, 3, 3), (16, 1, 2), (17, 1, 1), (18, 1, 2), (19, 2, 1), (20, 1, 1),
                                                    (21, 1, 1)])
def test_lde_matches_oracle(bpg, oracle, log_n, rate_bits, n_cols, ntt_form):
    rng = np.random.default_rng(200 + log_n)
    vals = rand_field(rng, (n_cols, 1 << log_n))
    want_coeffs, want_lde = oracle.lde_batch(vals, rate_bits)
    coeffs, lde = bpg.ops.lde_batch(to_dev(vals), rate_bits)
    br = bitrev_perm(log_n)
    assert (to_host(coeffs)[:, br] == want_coeffs).all()
    idx = coset_major_to_natural(log_n, rate_bits)
    assert (to_host(lde)[:, idx] == want_lde).all()
    # from_coeffs path (quotient chunks upstream)
    c2, lde2 = bpg.ops.lde_batch(coeffs, rate_bits, from_coeffs=True)
    assert (to_host(lde2) == to_host(lde)).all() and (to_host(c2) == to_host(coeffs)).all()


def test_persistent_lde_workgroups_give_the_one_shot_grid_s_values(bpg, oracle):
    """2^14-point coset-LDE blocks run as persistent workgroups that prefetch the next block's coefficients
    (ntt16_dit_persist_kernel) once a launch has more (block, coset) items than resident workgroups: with 8 / 16 / 32
    resident workgroups -- every workgroup walks over several items, padding ids included (37 columns: the last group of
    eight is padded) -- the values are those of the one-shot grid and of the oracle; inputs may be any u64."""
    import ctypes as C
    L = bpg.lib()
    L.bp_tune_ntt_persist.argtypes = [C.c_int, C.c_int]
    L.bp_tune_ntt_persist.restype = None
    rng = np.random.default_rng(77)
    log_n, n_cols = 14, 37
    vals = rng.integers(0, 1 << 64, size=(n_cols, 1 << log_n), dtype=np.uint64)
    vals[:, ::5] = np.uint64(2**64 - 1)
    try:
        L.bp_tune_ntt_mx(0)
        for r in (1, 3):
            outs = []
            for on, wgs in ((0, 256), (1, 8), (1, 16), (1, 32)):
                L.bp_tune_ntt_persist(on, wgs)
                outs.append(to_host(bpg.ops.lde_batch(to_dev(vals), r)[1]))
            for k in range(1, len(outs)):
                assert (outs[k] == outs[0]).all(), "persistent form %d differs from the one-shot grid (rate bits %d)" % (k, r)
            _, want = oracle.lde_batch(vals[:2] % np.uint64(P), r)
            assert (outs[1][:2][:, coset_major_to_natural(log_n, r)] == want).all()
    finally:
        L.bp_tune_ntt_persist(0, 256)
        L.bp_tune_ntt_mx(3)


def test_field_ops_against_big_integers(bpg):
    """K1 (SURVEY.md section 8(c) self-consistency item 2): every device form of the modular multiply
    (one-element carry chain, groups of three and four, the compiler form), lazy add/sub, the
    unreduced dot-product accumulator, 7*x, the inverse and the extension product, against Python
    integers -- on random operands and on the edges, including NON-canonical u64 inputs."""
    rng = np.random.default_rng(11)
    edge = [0, 1, 2, 7, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, P - 1, P, P + 1, (1 << 63), (1 << 64) - 1,
            (1 << 64) - (1 << 32), (1 << 64) - (1 << 32) - 1, 0xFFFFFFFF00000000, 0x00000000FFFFFFFF,
            0xFFFFFFFEFFFFFFFF, 0x8000000080000000, P - (1 << 32), (1 << 48) + 12345]
    a = [x for x in edge for _ in edge] + [int(v) for v in rng.integers(0, 1 << 64, 4096, dtype=np.uint64)]
    b = [y for _ in edge for y in edge] + [int(v) for v in rng.integers(0, 1 << 64, 4096, dtype=np.uint64)]
    # a block of all-ones operands drives the accumulator's wrap counters
    a += [(1 << 64) - 1] * 8
    b += [(1 << 64) - 1] * 8
    n = len(a)
    av, bv = np.array(a, dtype=np.uint64), np.array(b, dtype=np.uint64)
    out = to_host(bpg.ops.field_ops(to_dev(av), to_dev(bv)))
    got = [[int(v) for v in row] for row in out]
    for i in range(n):
        x, y = a[i], b[i]
        prod = x * y % P
        assert got[0][i] == prod and got[1][i] == prod and got[2][i] == prod and got[3][i] == prod, (i, hex(x), hex(y))
        assert got[4][i] == (x + y) % P and got[5][i] == (x - y) % P, (i, hex(x), hex(y))
        y0 = b[i - i % 4]
        assert got[6][i] == (2 * x * y + x * y0) % P, (i, hex(x), hex(y), hex(y0))
        assert got[7][i] == 7 * x % P
        assert got[8][i] == (pow(x % P, P - 2, P) if x % P else 0)
        e0, e1, f0, f1 = x % P, y % P, y % P, (x ^ y) % P
        assert got[9][i] == (e0 * f0 + 7 * e1 * f1) % P and got[10][i] == (e0 * f1 + e1 * f0) % P
        # group forms (gl::add_n / sub_n / canon_n): +-EPS under a mask in two instructions, on every edge pair
        assert got[11][i] == (x + y) % P and got[12][i] == (x - y) % P, (i, hex(x), hex(y))
        assert got[13][i] == x % P and got[14][i] == y % P, (i, hex(x), hex(y))


@pytest.fixture(params=["mx4", "mx4-ungrouped", "mx4-2groups", "mx2", "mx1", "lane"])
def perm_form(request, bpg):
    """the forms of the batch permutation: MDS on the matrix cores with 4 / 2 / 1 sets of 16 states per wave (four
    sets: all 22 partial rounds in three groups, rounds 4..19 in two groups of eight, or every round by itself), and
    one lane per state"""
    bpg.lib().bp_tune_poseidon_mx(0 if request.param == "lane" else 1)
    bpg.lib().bp_tune_poseidon_mx_sets(int(request.param[2:3]) if request.param != "lane" else 0)
    bpg.lib().bp_tune_poseidon_grouped(0 if request.param.endswith("-ungrouped") else 2 if request.param.endswith("-2groups") else 3)
    yield request.param
    bpg.lib().bp_tune_poseidon_mx(1)
    bpg.lib().bp_tune_poseidon_mx_sets(0)
    bpg.lib().bp_tune_poseidon_grouped(3)


def test_poseidon_kat_and_random(bpg, oracle, perm_form):
    rng = np.random.default_rng(7)
    states = rand_field(rng, (4099, 12))
    states[0] = 0
    states[1] = np.arange(12)
    states[2] = P - 1
    got = to_host(bpg.ops.poseidon_perm_batch_(to_dev(states)))
    assert [int(x) for x in got[0][:4]] == [0x3c18a9786cb0b359, 0xc4055e3364a246c3, 0x7953db0ab48808f4,
                                           0xc71603f33a1144ca]
    assert int(got[1][0]) == 0xd64e1e3efc5b8e9e
    assert (got == oracle.poseidon(states)).all()


def test_poseidon_noncanonical_inputs(bpg, oracle, perm_form):
    # inputs in [p, 2^64) must behave as their residues
    s = np.full((64, 12), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    s[1] = P
    s[2] = P + 5
    s[3] = 0xFFFFFFFF                      # every byte plane of the low half at its maximum
    s[4] = 0xFFFFFFFF00000000
    s[5, ::2] = P - 1
    got = to_host(bpg.ops.poseidon_perm_batch_(to_dev(s)))
    assert (got == oracle.poseidon(s % np.uint64(P))).all()


def test_poseidon_byte_plane_extremes(bpg, oracle, perm_form):
    # the matrix-core MDS sums byte planes: states whose words are all-0xFF / all-0x00 / 0x80 / 0x7F bytes drive every
    # plane sum to its bounds (and the signed-byte offset to both ends); ragged batch sizes cover the clamped tail
    pats = [0, 0xFFFFFFFFFFFFFFFF % P, 0x8080808080808080, 0x7F7F7F7F7F7F7F7F, 0xFF00FF00FF00FF00, 0x00FF00FF00FF00FF,
            0x0101010101010101, 0xFEFEFEFEFEFEFEFE]
    rng = np.random.default_rng(11)
    for n in (1, 15, 16, 17, 63, 64, 65, 255, 257):
        s = np.array([[pats[(i + k * (i % 3 + 1)) % len(pats)] for k in range(12)] for i in range(n)], dtype=np.uint64)
        s[n // 2] = rand_field(rng, (12,))
        got = to_host(bpg.ops.poseidon_perm_batch_(to_dev(s.copy())))
        assert (got == oracle.poseidon(s % np.uint64(P))).all(), n


@pytest.mark.parametrize("quad", ["quad", "lane", "mx4", "mx2", "mx1", "mx", "mx+fused", "mx+fused+wide", "quad+fused",
                                  "mx4-ungrouped", "mx4-2groups"])
@pytest.mark.parametrize("log_n,rate_bits,n_cols,cap_h", [(3, 1, 3, 4), (4, 1, 4, 0), (6, 1, 8, 4), (7, 3, 19, 4),
                                                          (10, 1, 135, 4), (12, 1, 33, 2), (9, 3, 2, 4), (5, 1, 9, 1),
                                                          (6, 1, 13, 3)])
def test_merkle_commit_matches_oracle(bpg, oracle, log_n, rate_bits, n_cols, cap_h, quad):
    # the three Poseidon kernel families: 4 lanes per state with DPP exchange / one lane per state / MDS on the
    # matrix cores (four sets of 16 states per wave)
    # "+fused": up to seven levels of at most 4096 nodes per launch (LDS hand-down), in the matrix-core one-set form or
    # the quad form; "mx4-ungrouped" / "mx4-2groups": four sets per wave with every partial round by itself / only rounds
    # 4..19 grouped (the default groups all 22)
    bpg.lib().bp_tune_quad_threshold((1 << 40) if quad.startswith("quad") else 1)  # 1: never quad; 0 would be automatic
    bpg.lib().bp_tune_poseidon_mx(1 if quad.startswith("mx") else 0)
    bpg.lib().bp_tune_poseidon_mx_sets(int(quad[2:3]) if quad[:3] in ("mx4", "mx2", "mx1") else 0)  # "mx": sets by launch size
    bpg.lib().bp_tune_merkle_fused(1 if "+fused" in quad else 0)
    bpg.lib().bp_tune_merkle_wide(14 if quad.endswith("+wide") else 0)   # nine levels per launch from 256 parents up
    bpg.lib().bp_tune_poseidon_grouped(0 if quad.endswith("-ungrouped") else 2 if quad.endswith("-2groups") else 3)
    if quad in ("mx", "mx+fused", "mx+fused+wide"):
        bpg.lib().bp_tune_quad_threshold(1 << (log_n + rate_bits))  # leaves with 4 sets, then 2, then 1 up the tree
    rng = np.random.default_rng(300 + log_n)
    rows = 1 << (log_n + rate_bits)
    lde_cm = rand_field(rng, (n_cols, rows))          # coset-major, as the LDE kernel writes it
    idx = coset_major_to_natural(log_n, rate_bits)
    lde_nat = np.ascontiguousarray(lde_cm[:, idx])    # natural order for the oracle
    want_dig, want_cap = oracle.merkle_commit(lde_nat, cap_h, bitrev_rows=True)
    dig = to_host(bpg.ops.merkle_commit(to_dev(lde_cm), log_n, rate_bits, cap_h))
    bpg.lib().bp_tune_quad_threshold(0)  # back to automatic
    bpg.lib().bp_tune_poseidon_mx(1)
    bpg.lib().bp_tune_poseidon_mx_sets(0)
    bpg.lib().bp_tune_merkle_fused(0)
    bpg.lib().bp_tune_merkle_wide(0)
    bpg.lib().bp_tune_poseidon_grouped(3)
    assert (dig == want_dig).all()
    assert (dig[-(1 << cap_h):] == want_cap).all()


# ---- the per-stage L0 entry points of SURVEY.md section 8(b): quotient, FRI fold, proof of work ----

@pytest.mark.parametrize("log_n,n_cols,n_const,deg_pow,rate_bits", [(6, 16, 0, 1, 1), (9, 24, 0, 1, 1), (8, 19, 5, 3, 3),
                                                                      (12, 136, 0, 1, 1), (10, 135, 7, 3, 3)])
def test_quotient_eval_matches_oracle(bpg, oracle, log_n, n_cols, n_const, deg_pow, rate_bits):
    """K5 alone: random LDE matrices (the constraints are evaluated on whatever is there), fixed challenges."""
    rng = np.random.default_rng(300 + log_n)
    n, rows = 1 << log_n, (1 << log_n) << rate_bits
    trace = rand_field(rng, (n_cols, rows))
    aux = rand_field(rng, (n_cols // 8, rows))
    consts = rand_field(rng, (n_const, rows)) if n_const else None
    ctl = rand_field(rng, (4,))
    alphas = rand_field(rng, (2,))
    want = oracle.quotient_values(oracle.make_cfg(log_n, n_cols, n_const=n_const, deg_pow=deg_pow, rate_bits=rate_bits),
                                  consts, trace, aux, ctl, alphas[0], alphas[1])
    # the device layout is coset-major: device position idx[i] holds natural point i
    idx = coset_major_to_natural(log_n, rate_bits)

    def to_cm(mat):
        cm = np.empty_like(mat)
        cm[:, idx] = mat
        return to_dev(cm)
    got = bpg.ops.quotient_eval(bpg.ops.stark_cfg(log_n, n_cols, n_const=n_const, deg_pow=deg_pow, rate_bits=rate_bits),
                                to_cm(trace), to_cm(aux), to_cm(consts) if n_const else None, ctl, alphas)
    assert (to_host(got)[:, idx] == want).all()


def test_quotient_scratch_fits_whatever_the_load_state_is_at_launch(bpg, oracle):
    """bp_quotient_scratch_words and bp_quotient_eval are two calls, and the unit spreading (hence the size of the
    partial sums) follows the device's load, which other threads change in between: the scratch size must cover both
    states.  The buffer is allocated at exactly the size returned and followed by a guard that must stay untouched."""
    import ctypes as C
    import torch
    log_n, n_cols, rate_bits = 9, 128, 1
    rng = np.random.default_rng(77)
    rows = (1 << log_n) << rate_bits
    trace, aux = rand_field(rng, (n_cols, rows)), rand_field(rng, (n_cols // 8, rows))
    ctl, alphas = rand_field(rng, (4,)), rand_field(rng, (2,))
    cfg = bpg.ops.stark_cfg(log_n, n_cols, rate_bits=rate_bits)
    L = bpg.lib()
    want = None
    try:
        for at_size, at_launch in ((1, 0), (0, 1), (0, 0), (1, 1)):
            L.bp_tune_assume_loaded(at_size)
            words = int(L.bp_quotient_scratch_words(0, C.byref(cfg)))
            guard = 4096
            buf = torch.full((words + guard,), 0x5A5A5A5A5A5A5A5A, dtype=torch.int64, device="cuda")
            out = torch.empty((2, rows), dtype=torch.int64, device="cuda")
            L.bp_tune_assume_loaded(at_launch)
            t, a = to_dev(trace), to_dev(aux)
            bpg._lib.check(L.bp_quotient_eval(0, C.byref(cfg), t.data_ptr(), a.data_ptr(), None,
                                              (C.c_uint64 * 4)(*[int(x) for x in ctl]), (C.c_uint64 * 2)(*[int(x) for x in alphas]),
                                              buf.data_ptr(), out.data_ptr(), None))
            torch.cuda.synchronize()
            assert (buf[words:] == 0x5A5A5A5A5A5A5A5A).all(), "the launch wrote past the scratch it was sized"
            got = to_host(out)
            if want is None:
                want = got
            assert (got == want).all()
    finally:
        L.bp_tune_assume_loaded(-1)


def test_merkle_commit_of_a_tiny_matrix_with_forced_sets(bpg, oracle):
    """bp_tune_poseidon_mx_sets(4) on a 16-row matrix: the four-set kernel would read 48 rows past the last column;
    the launcher takes fewer sets for trees smaller than a wave's sets.  Same digests in every form."""
    rng = np.random.default_rng(78)
    L = bpg.lib()
    for log_n, r in ((3, 1), (4, 0), (5, 0)):
        rows = (1 << log_n) << r
        lde = rand_field(rng, (9, rows))
        ref = None
        try:
            for sets in (0, 1, 2, 4):
                L.bp_tune_poseidon_mx_sets(sets)
                got = to_host(bpg.ops.merkle_commit(to_dev(lde), log_n, r, 2))
                ref = got if ref is None else ref
                assert (got == ref).all()
        finally:
            L.bp_tune_poseidon_mx_sets(0)


@pytest.mark.parametrize("log_nl,rate_bits", [(4, 1), (6, 1), (9, 3), (13, 3), (16, 1)])
def test_fri_fold_matches_oracle(bpg, oracle, log_nl, rate_bits):
    """K6 alone: one arity-16 fold of random extension values, device coset-major vs oracle bit-reversed."""
    rng = np.random.default_rng(400 + log_nl)
    log_m = log_nl + rate_bits
    m = 1 << log_m
    vals = rand_field(rng, (m, 2))                      # natural order: index i <-> shift * w_m^i
    beta = rand_field(rng, (2,))
    shift = int(pow(7, 16, P))                          # a later layer's domain shift
    br = bitrev_perm(log_m)
    want_br = oracle.fri_fold(vals[br], 4, shift, beta)  # [m/16, 2] in bit-reversed order
    idx = coset_major_to_natural(log_nl, rate_bits)       # coset-major position of each natural index
    cm = np.empty_like(vals)
    cm[idx] = vals
    got = to_host(bpg.ops.fri_fold(to_dev(cm), log_nl, rate_bits, shift, beta))
    nat = got[coset_major_to_natural(log_nl - 4, rate_bits)]   # natural order of the folded domain
    assert (nat[bitrev_perm(log_m - 4)] == want_br).all()


def test_pow_grind_smallest_witness(bpg, oracle):
    """K9 alone: the smallest nonce, against an exhaustive search with the oracle's permutation."""
    rng = np.random.default_rng(500)
    for bits, pos in ((6, 0), (10, 3), (12, 7)):
        state = rand_field(rng, (12,))
        got = bpg.ops.pow_grind(state, pos, bits)
        tries = np.tile(state, (got + 1, 1))
        tries[:, pos] = np.arange(got + 1, dtype=np.uint64)
        out = oracle.poseidon(tries)
        ok = (out[:, 7] >> np.uint64(64 - bits)) == 0
        assert ok[got] and not ok[:got].any()


@pytest.mark.parametrize("log_n,n_cols", [(0, 2), (3, 3), (8, 5), (10, 2)])
def test_openings_match_horner(bpg, log_n, n_cols):
    """K8 alone: every column polynomial at two extension points, against Horner's rule on Python integers
    in F_p[X]/(X^2 - 7)."""
    rng = np.random.default_rng(600 + log_n)
    n = 1 << log_n
    coeffs = rand_field(rng, (n_cols, n))               # natural order
    z0, z1 = rand_field(rng, (2,)), rand_field(rng, (2,))

    def horner(c, z):
        a0 = a1 = 0
        for v in reversed([int(x) for x in c]):
            a0, a1 = (a0 * z[0] + 7 * a1 * z[1] + v) % P, (a0 * z[1] + a1 * z[0]) % P
        return a0, a1
    got = to_host(bpg.ops.openings(to_dev(np.ascontiguousarray(coeffs[:, bitrev_perm(log_n)])), z0, z1))
    for c in range(n_cols):
        w0, w1 = horner(coeffs[c], [int(z0[0]), int(z0[1])]), horner(coeffs[c], [int(z1[0]), int(z1[1])])
        assert tuple(int(v) for v in got[c]) == w0 + w1
    one = to_host(bpg.ops.openings(to_dev(np.ascontiguousarray(coeffs[:, bitrev_perm(log_n)])), z0))
    assert (one[:, :2] == got[:, :2]).all() and (one[:, 2:] == 0).all()
