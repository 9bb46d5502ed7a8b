"""Size-independent properties at BASELINE sizes (where running the oracle per test is too slow):
round trips, linearity, coset structure of the LDE, Merkle paths against caps, determinism."""
import numpy as np
import pytest

from util import P, bitrev_perm, leaf_of_coset_major, rand_field, to_dev, to_host

pytestmark = pytest.mark.gpu


def fsub(a, b):
    return np.where(a >= b, a - b, a + (np.uint64(P) - b))


def fadd(a, b):
    s = a + b
    return np.where((s < a) | (s >= np.uint64(P)), s - np.uint64(P), s)


@pytest.mark.parametrize("log_n,n_cols", [(14, 64), (16, 16), (18, 3), (20, 4), (22, 2), (24, 2)])
def test_ntt_round_trip_and_linearity(bpg, log_n, n_cols):
    rng = np.random.default_rng(log_n)
    a = rand_field(rng, (n_cols, 1 << log_n))
    b = rand_field(rng, (n_cols, 1 << log_n))
    ca = to_host(bpg.ops.ntt_batch_(to_dev(a), bpg.ops.NTT_INV_NAT2BR))
    back = to_host(bpg.ops.ntt_batch_(to_dev(ca), bpg.ops.NTT_FWD_BR2NAT))
    assert (back == a).all()                                        # fft(ifft(x)) == x
    cb = to_host(bpg.ops.ntt_batch_(to_dev(b), bpg.ops.NTT_INV_NAT2BR))
    cab = to_host(bpg.ops.ntt_batch_(to_dev(fadd(a, b)), bpg.ops.NTT_INV_NAT2BR))
    assert (cab == fadd(ca, cb)).all()                              # linear
    # a constant column is the constant polynomial: only coefficient 0 (position 0 either order)
    const = np.full((1, 1 << log_n), 12345, dtype=np.uint64)
    cc = to_host(bpg.ops.ntt_batch_(to_dev(const), bpg.ops.NTT_INV_NAT2BR))
    assert int(cc[0, 0]) == 12345 and not cc[0, 1:].any()


@pytest.mark.parametrize("log_n,rate_bits,n_cols", [(14, 1, 32), (13, 3, 16), (17, 1, 4), (20, 1, 2)])
def test_lde_structure(bpg, log_n, rate_bits, n_cols):
    """Coset t of the LDE is the evaluation on 7*w^t*<w_n>: interpolating any single coset back must
    give the same coefficients, and a degree-0 polynomial is constant on every coset."""
    rng = np.random.default_rng(50 + log_n)
    n = 1 << log_n
    vals = rand_field(rng, (n_cols, n))
    coeffs, lde = bpg.ops.lde_batch(to_dev(vals), rate_bits)
    c2, lde2 = bpg.ops.lde_batch(coeffs, rate_bits, from_coeffs=True)
    assert (to_host(lde2) == to_host(lde)).all()
    # the interpolant reproduces the trace on the subgroup itself
    back = to_host(bpg.ops.ntt_batch_(coeffs.clone(), bpg.ops.NTT_FWD_BR2NAT))
    assert (back == vals).all()
    const = np.full((1, n), 777, dtype=np.uint64)
    _, lc = bpg.ops.lde_batch(to_dev(const), rate_bits)
    assert (to_host(lc) == 777).all()
    # x -> X (the identity polynomial): LDE values are the domain points themselves, 7*w_M^(t + 2^r m)
    w_n = pow(7, (P - 1) >> log_n, P)
    xs = np.array([pow(w_n, i, P) for i in range(min(n, 4096))], dtype=np.uint64)
    if n <= 4096:
        _, lx = bpg.ops.lde_batch(to_dev(xs[None, :]), rate_bits)
        lx = to_host(lx)[0]
        w_m = pow(7, (P - 1) >> (log_n + rate_bits), P)
        for t in range(1 << rate_bits):
            for m in (0, 1, n - 1):
                assert int(lx[t * n + m]) == 7 * pow(w_m, t + (m << rate_bits), P) % P


@pytest.mark.parametrize("log_n,rate_bits,n_cols", [(16, 1, 9), (19, 1, 3), (13, 3, 135)])
def test_merkle_paths_verify_against_cap_at_size(bpg, oracle, log_n, rate_bits, n_cols):
    rng = np.random.default_rng(90 + log_n)
    rows = 1 << (log_n + rate_bits)
    lde = rand_field(rng, (n_cols, rows))
    dig = to_host(bpg.ops.merkle_commit(to_dev(lde), log_n, rate_bits, 4))
    cap = np.ascontiguousarray(dig[-16:].reshape(-1))
    leaf_of = leaf_of_coset_major(log_n, rate_bits)
    log_l = log_n + rate_bits
    L = oracle.lib()
    flat = np.ascontiguousarray(dig.reshape(-1))
    for pos in rng.integers(0, rows, size=12):
        leaf = int(leaf_of[pos])
        path = np.empty((log_l - 4) * 4, dtype=np.uint64)
        L.orc_merkle_path(flat, log_l, 4, leaf, path)               # sibling walk in the GPU's digest buffer
        row = np.ascontiguousarray(lde[:, pos])
        assert L.orc_merkle_verify(row, n_cols, leaf, path, log_l, 4, cap) == 0
        row[0] ^= np.uint64(1)
        assert L.orc_merkle_verify(row, n_cols, leaf, path, log_l, 4, cap) != 0


def test_poseidon_batch_is_deterministic_and_elementwise(bpg, oracle):
    rng = np.random.default_rng(3)
    s = rand_field(rng, (1 << 18, 12))
    a = to_host(bpg.ops.poseidon_perm_batch_(to_dev(s)))
    b = to_host(bpg.ops.poseidon_perm_batch_(to_dev(s[::-1].copy())))[::-1]
    assert (a == b).all()
    idx = rng.integers(0, s.shape[0], size=64)
    assert (a[idx] == oracle.poseidon(s[idx])).all()


def test_same_inputs_same_proof_bytes(bpg):
    cfg = bpg.ops.stark_cfg(12, 40, num_queries=20, pow_bits=10)
    p1 = bpg.ops.stark_prove_synthetic(cfg, 99)
    p2 = bpg.ops.stark_prove_synthetic(cfg, 99)
    p3 = bpg.ops.stark_prove_synthetic(cfg, 100)
    assert (p1 == p2).all() and (p1 != p3).any()


def test_every_kernel_form_gives_the_same_commitment_on_random_shapes(bpg):
    """Differential test of the Poseidon kernel families on shapes the oracle-backed tests do not list: the
    matrix-core forms (4 / 2 / 1 sets of 16 states per wave, and the size-dependent mix), one lane per state and the
    quad-cooperative kernels must produce the same digest buffer for ragged widths (absorb tails of 1..7 words,
    hash_or_noop rows of <= 4 columns), leaf counts below one wave and cap heights up to the leaf level."""
    rng = np.random.default_rng(20261004)
    L = bpg.lib()
    shapes = [(int(rng.integers(0, 12)), int(rng.integers(0, 3)), int(rng.integers(1, 41))) for _ in range(40)]
    shapes += [(0, 0, 1), (0, 0, 9), (1, 0, 5), (2, 1, 4), (13, 1, 23), (14, 1, 8), (11, 3, 17)]
    try:
        for log_n, rate_bits, n_cols in shapes:
            rows = 1 << (log_n + rate_bits)
            cap_h = int(rng.integers(0, min(4, log_n + rate_bits) + 1))
            lde = to_dev(rand_field(rng, (n_cols, rows)))
            got = {}
            for form, (mx, sets, thr) in {"lane": (0, 0, 1), "quad": (0, 0, 1 << 40), "mx4": (1, 4, 1), "mx2": (1, 2, 1),
                                          "mx1": (1, 1, 1), "mx": (1, 0, max(2, rows // 2))}.items():
                L.bp_tune_poseidon_mx(mx)
                L.bp_tune_poseidon_mx_sets(sets)
                L.bp_tune_quad_threshold(thr)
                got[form] = to_host(bpg.ops.merkle_commit(lde, log_n, rate_bits, cap_h))
            for form, dig in got.items():
                assert (dig == got["lane"]).all(), (form, log_n, rate_bits, n_cols, cap_h)
    finally:
        L.bp_tune_poseidon_mx(1)
        L.bp_tune_poseidon_mx_sets(0)
        L.bp_tune_quad_threshold(0)


def test_permutation_forms_agree_on_a_large_random_batch(bpg):
    rng = np.random.default_rng(77)
    L = bpg.lib()
    s = rng.integers(0, 1 << 64, (100003, 12), dtype=np.uint64)   # non-canonical words included, ragged tail
    try:
        outs = []
        for mx, sets in ((0, 0), (1, 4), (1, 2), (1, 1)):
            L.bp_tune_poseidon_mx(mx)
            L.bp_tune_poseidon_mx_sets(sets)
            outs.append(to_host(bpg.ops.poseidon_perm_batch_(to_dev(s.copy()))))
        for o in outs[1:]:
            assert (o == outs[0]).all()
        assert (outs[0] < np.uint64(P)).all()
    finally:
        L.bp_tune_poseidon_mx(1)
        L.bp_tune_poseidon_mx_sets(0)
