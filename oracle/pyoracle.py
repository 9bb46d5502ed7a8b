"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(the product package must never do so).  "parity unpinned" by the reference: see oracle/gl.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")


def build(force=False):
    """make decides what is stale (the Makefile lists every source and include of liboracle.so): no second list here
    that an added file could be missing from."""
    subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []) + ["liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


class StarkCfg(C.Structure):
    _fields_ = ([(n, C.c_uint32) for n in ("log_n", "n_cols", "n_const", "deg_pow", "rate_bits", "cap_height",
                                            "num_queries", "pow_bits", "arity_bits", "final_poly_bits", "air_id")]
                + [("pub", C.c_uint64 * 4)])   # the table's public inputs (AIR 8); zero otherwise


class Challenger(C.Structure):
    _fields_ = [("state", C.c_uint64 * 12), ("inb", C.c_uint64 * 8), ("n_in", C.c_uint),
                ("outb", C.c_uint64 * 8), ("n_out", C.c_uint)]


class PgConfig(C.Structure):
    """orc_pg_config: the proving parameters of bp_config without the device/worker fields."""
    _fields_ = ([("table_log_lo", C.c_uint32 * 7), ("table_log_hi", C.c_uint32 * 7)]
                + [(n, C.c_uint32) for n in ("stark_rate_bits", "stark_cap_height", "stark_num_queries",
                                             "stark_pow_bits", "arity_bits", "final_poly_bits", "rec_log_n",
                                             "rec_n_cols", "rec_n_const", "rec_rate_bits", "rec_num_queries",
                                             "rec_pow_bits", "shrink_depth", "rec_air_id")])


class Gl2(C.Structure):
    _fields_ = [("c0", C.c_uint64), ("c1", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp, u, sz, i, u6 = C.c_void_p, C.c_uint, C.c_size_t, C.c_int, C.c_uint64
    L.orc_dft_naive.argtypes = [u64p, u64p, u, i]
    L.orc_ntt.argtypes = [u64p, u]
    L.orc_intt.argtypes = [u64p, u]
    L.orc_coset_ntt.argtypes = [u64p, u, u6]
    L.orc_coset_intt.argtypes = [u64p, u, u6]
    L.orc_ntt_batch.argtypes = [u64p, u, sz, sz, i]
    L.orc_lde_batch.argtypes = [u64p, vp, u64p, u, u, sz, i]
    L.orc_poseidon_batch.argtypes = [u64p, sz]
    L.orc_hash_no_pad.argtypes = [u64p, sz, u64p]
    L.orc_hash_or_noop.argtypes = [u64p, sz, u64p]
    L.orc_merkle_digest_words.argtypes = [u, u]
    L.orc_merkle_digest_words.restype = sz
    L.orc_merkle_commit.argtypes = [u64p, sz, sz, u, u, i, u64p]
    L.orc_merkle_commit_rows.argtypes = [u64p, sz, u, u, u64p]
    L.orc_merkle_path.argtypes = [u64p, u, u, sz, u64p]
    L.orc_merkle_verify.argtypes = [u64p, sz, sz, u64p, u, u, u64p]
    L.orc_ch_init.argtypes = [C.POINTER(Challenger)]
    L.orc_ch_observe.argtypes = [C.POINTER(Challenger), u6]
    L.orc_ch_observe_many.argtypes = [C.POINTER(Challenger), u64p, sz]
    L.orc_ch_challenge.argtypes = [C.POINTER(Challenger)]
    L.orc_ch_challenge.restype = u6
    L.orc_fri_fold.argtypes = [u64p, u64p, u, u, u6, Gl2]
    cfgp = C.POINTER(StarkCfg)
    for name in ("orc_cfg_n_aux", "orc_cfg_n_quot", "orc_cfg_n_layers"):
        getattr(L, name).argtypes = [cfgp]
        getattr(L, name).restype = C.c_uint32
    L.orc_proof_words.argtypes = [cfgp]
    L.orc_proof_words.restype = sz
    L.orc_synth_constants.argtypes = [u6, u, sz, u64p]
    L.orc_synth_trace.argtypes = [u6, cfgp, vp, u64p]
    L.orc_keccak_f.argtypes = [u64p]
    L.orc_keccak_trace.argtypes = [u6, vp, u, u64p]
    L.orc_logic_trace.argtypes = [u6, vp, u, u64p]
    L.orc_memory_trace.argtypes = [u6, vp, u, u64p]
    L.orc_arithmetic_trace.argtypes = [u6, vp, u, u64p]
    L.orc_byte_packing_trace.argtypes = [u6, vp, u, u64p]
    L.orc_keccak_sponge_trace.argtypes = [u6, vp, u, u64p]
    L.orc_arithmetic_mul_trace.argtypes = [u6, vp, u, u64p]
    L.orc_keccak_sponge_rows.argtypes = [C.c_char_p, sz, vp, vp]
    L.orc_keccak_sponge_rows.restype = sz
    L.orc_commit_values.argtypes = [u64p, u, sz, u, u]
    L.orc_commit_values.restype = vp
    L.orc_commit_coeffs.argtypes = [u64p, u, sz, u, u]
    L.orc_commit_coeffs.restype = vp
    for name in ("orc_committed_cap", "orc_committed_lde", "orc_committed_coeffs", "orc_committed_digests"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = C.POINTER(C.c_uint64)
    L.orc_committed_free.argtypes = [vp]
    L.orc_stark_prove.argtypes = [cfgp, vp, vp, u64p, u64p, C.POINTER(Challenger), u64p]
    L.orc_quotient_values.argtypes = [cfgp, vp, u64p, u64p, u64p, u6, u6, u64p]
    L.orc_stark_verify.argtypes = [cfgp, vp, u64p, C.POINTER(Challenger), u64p]
    L.orc_proof_digest.argtypes = [cfgp, u64p, u64p]
    L.orc_pg_state_build.argtypes = [C.POINTER(PgConfig)]
    L.orc_pg_state_build.restype = vp
    L.orc_pg_state_free.argtypes = [vp]
    L.orc_pg_txn.argtypes = [vp, u64p, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(sz)]
    L.orc_pg_txn_keccak.argtypes = [vp, u64p, vp, sz, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(sz)]
    L.orc_pg_txn_witness.argtypes = [vp, u64p, C.POINTER(vp), C.POINTER(sz), C.POINTER(C.c_int),
                                     C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(sz)]
    L.orc_stark_public_inputs.argtypes = [u6, u64p]
    L.orc_plonk_constants.argtypes = [u6, u, u, u, u, u, u, u64p]
    L.orc_plonk_trace.argtypes = [u6, u64p, u, u, u, u, u, u64p, u64p, u, u64p]
    L.orc_stark_public_input_list.argtypes = [u6, u64p]
    L.orc_pg_txn_tables.argtypes = [vp, u64p, C.POINTER(vp), C.POINTER(sz), C.POINTER(C.c_int),
                                    C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(sz)]
    L.orc_pg_verify_tables.argtypes = [C.POINTER(PgConfig), u64p, sz]
    L.orc_ctl_n_aux.argtypes = [u, u]
    L.orc_ctl_n_aux.restype = C.c_uint32
    L.orc_pg_preprocess.argtypes = [vp, u64p]
    L.orc_pg_agg.argtypes = [vp, u64p, sz, i, u64p, sz, i, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(sz)]
    L.orc_pg_block.argtypes = [vp, vp, sz, u64p, sz, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(sz)]
    L.orc_pg_verify.argtypes = [vp, u64p, sz]
    L.orc_free.argtypes = [vp]
    L.orc_pg_circuit_cap.argtypes = [vp, i, u64p]
    _lib = L
    return L


P = 0xFFFFFFFF00000001


def arr(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


def ntt(a, inverse=False):
    a = arr(a).copy()
    log_n = int(a.shape[-1]).bit_length() - 1
    (lib().orc_intt if inverse else lib().orc_ntt)(a, log_n)
    return a


def dft_naive(a, inverse=False):
    a = arr(a)
    out = np.empty_like(a)
    lib().orc_dft_naive(a, out, int(a.size).bit_length() - 1, int(inverse))
    return out


def ntt_batch(cols, inverse=False):
    cols = arr(cols).copy()
    n_cols, n = cols.shape
    lib().orc_ntt_batch(cols, n.bit_length() - 1, n_cols, n, int(inverse))
    return cols


def lde_batch(values, rate_bits, from_coeffs=False):
    values = arr(values)
    n_cols, n = values.shape
    coeffs = np.empty_like(values)
    lde = np.empty((n_cols, n << rate_bits), dtype=np.uint64)
    lib().orc_lde_batch(values, coeffs.ctypes.data, lde, n.bit_length() - 1, rate_bits, n_cols, int(from_coeffs))
    return coeffs, lde


def poseidon(states):
    s = arr(states).copy().reshape(-1, 12)
    lib().orc_poseidon_batch(s, s.shape[0])
    return s


def hash_no_pad(x):
    x = arr(x)
    out = np.empty(4, dtype=np.uint64)
    lib().orc_hash_no_pad(x, x.size, out)
    return out


def hash_or_noop(x):
    x = arr(x)
    out = np.empty(4, dtype=np.uint64)
    lib().orc_hash_or_noop(x, x.size, out)
    return out


def merkle_commit(cols, cap_h, bitrev_rows=True):
    """cols: [n_cols, n_leaves] column-major.  Returns (digests[levels...,4], cap[2^cap_h,4])."""
    cols = arr(cols)
    n_cols, n = cols.shape
    log_l = n.bit_length() - 1
    words = lib().orc_merkle_digest_words(log_l, cap_h)
    dig = np.empty(words, dtype=np.uint64)
    lib().orc_merkle_commit(cols, n, n_cols, log_l, cap_h, int(bitrev_rows), dig)
    return dig.reshape(-1, 4), dig[-(4 << cap_h):].reshape(-1, 4).copy()


def merkle_commit_rows(leaves, cap_h):
    leaves = arr(leaves)
    n, leaf_len = leaves.shape
    log_l = n.bit_length() - 1
    dig = np.empty(lib().orc_merkle_digest_words(log_l, cap_h), dtype=np.uint64)
    lib().orc_merkle_commit_rows(leaves, leaf_len, log_l, cap_h, dig)
    return dig.reshape(-1, 4), dig[-(4 << cap_h):].reshape(-1, 4).copy()


def fri_fold(values, arity_bits, shift, beta):
    """values: [m, 2] ext elements, bit-reversed order."""
    v = arr(values)
    m = v.shape[0]
    out = np.empty((m >> arity_bits, 2), dtype=np.uint64)
    lib().orc_fri_fold(v, out, m.bit_length() - 1, arity_bits, shift, Gl2(int(beta[0]), int(beta[1])))
    return out


class PyChallenger:
    def __init__(self):
        self.c = Challenger()
        lib().orc_ch_init(C.byref(self.c))

    def observe(self, xs):
        xs = arr(np.atleast_1d(xs)).ravel()
        lib().orc_ch_observe_many(C.byref(self.c), xs, xs.size)

    def challenge(self):
        return int(lib().orc_ch_challenge(C.byref(self.c)))

    def clone(self):
        o = PyChallenger()
        C.memmove(C.byref(o.c), C.byref(self.c), C.sizeof(Challenger))
        return o


AIR_SYNTHETIC, AIR_KECCAK_F, AIR_LOGIC, AIR_MEMORY, AIR_ARITHMETIC, AIR_BYTE_PACKING = 0, 1, 2, 3, 4, 5
AIR_KECCAK_SPONGE = 6
AIR_ARITHMETIC_MUL = 7
KECCAK_COLS = 2431
LOGIC_COLS = 524
MEMORY_COLS = 45
ARITHMETIC_COLS = 309
BYTE_PACKING_COLS = 299
KECCAK_SPONGE_COLS = 2414
ARITHMETIC_MUL_COLS = 1217


def make_cfg(log_n, n_cols, n_const=0, deg_pow=1, rate_bits=1, cap_height=4, num_queries=84, pow_bits=16,
             arity_bits=4, final_poly_bits=5, air_id=AIR_SYNTHETIC, pub=(0, 0, 0, 0)):
    return StarkCfg(log_n, n_cols, n_const, deg_pow, rate_bits, cap_height, num_queries, pow_bits, arity_bits,
                    final_poly_bits, air_id, (C.c_uint64 * 4)(*[int(x) for x in pub]))


AIR_PLONK, PLONK_COLS, PLONK_CONSTS = 8, 135, 85


def plonk_cfg(log_n, pub=(0, 0, 0, 0), **kw):
    """AIR 8 (plonk_air.c): 135 wires, 85 constant columns, degree 9 -> deg_pow 3, rate_bits 3."""
    return make_cfg(log_n, PLONK_COLS, n_const=PLONK_CONSTS, deg_pow=3, rate_bits=3, air_id=AIR_PLONK, pub=pub, **kw)


def stark_public_inputs(seed):
    out = np.empty(4, dtype=np.uint64)
    lib().orc_stark_public_inputs(C.c_uint64(seed), out)
    return out


def stark_public_input_list(seed):
    """the four-word public-input list of a lone AIR-8 table proof; stark_public_inputs(seed) is its hash"""
    out = np.empty(4, dtype=np.uint64)
    lib().orc_stark_public_input_list(C.c_uint64(seed), out)
    return out


def plonk_constants(log_n, seed, pi_len=4, n_paths=0, depth=0, path_pi0=0, leaf_len=0):
    """the 85 constant columns of a circuit that hashes a public-input list of pi_len words and walks n_paths Merkle
    paths of `depth` levels (leaf digest / cap entry = list words path_pi0 + 8 p .. + 7); leaf_len > 0: it also hashes the
    leaf_len-word row each path's leaf digest is the hash of"""
    out = np.zeros((PLONK_CONSTS, 1 << log_n), dtype=np.uint64)
    lib().orc_plonk_constants(C.c_uint64(seed), log_n, pi_len, n_paths, depth, path_pi0, leaf_len, out)
    return out


def plonk_trace(log_n, seed, pi, consts, n_paths=0, depth=0, path_pi0=0, paths=None, leaf_len=0):
    """the witness; pi: the public-input list the hash rows absorb (its hash lands in row 0); paths: per Merkle path
    1 + 4 depth + leaf_len words (the leaf's position, the siblings upward, the row the leaf digest is the hash of)"""
    out = np.zeros((PLONK_COLS, 1 << log_n), dtype=np.uint64)
    pi = arr(pi)
    pw = arr(paths) if n_paths else np.zeros(1, dtype=np.uint64)
    assert not n_paths or pw.size == n_paths * (1 + 4 * depth + leaf_len)
    lib().orc_plonk_trace(C.c_uint64(seed), pi, pi.size, n_paths, depth, path_pi0, leaf_len, pw, arr(consts), log_n, out)
    return out


def merkle_path(leaf_digests, index, cap_height=0):
    """Sibling digests of leaf `index` in the tree over leaf_digests ([n][4], n a power of two), leaf upward, down to
    2^cap_height cap entries; and the cap entry the path arrives at.  Built with hash_no_pad over (left, right)."""
    level = [np.array(d, dtype=np.uint64) for d in leaf_digests]
    sibs, i = [], index
    while len(level) > (1 << cap_height):
        sibs.append(level[i ^ 1])
        level = [hash_no_pad(np.concatenate([level[2 * k], level[2 * k + 1]])) for k in range(len(level) // 2)]
        i >>= 1
    return np.concatenate(sibs) if sibs else np.zeros(0, dtype=np.uint64), level[i]


def keccak_f(lanes):
    """orc_keccak_f: the permutation on 25 lanes (index x + 5y)."""
    a = np.ascontiguousarray(lanes, dtype=np.uint64).copy()
    lib().orc_keccak_f(a)
    return a


def keccak_trace(log_n, seed=0, inputs=None):
    """orc_keccak_trace: the AIR-1 witness [2431, 2^log_n] (the last column, the lookup's filter, zero)."""
    out = np.zeros((KECCAK_COLS, 1 << log_n), dtype=np.uint64)
    inp = np.ascontiguousarray(inputs, dtype=np.uint64) if inputs is not None else None
    if inp is not None:
        assert inp.shape == (((1 << log_n) + 23) // 24, 25)
    lib().orc_keccak_trace(seed, inp.ctypes.data if inp is not None else None, log_n, out)
    return out


def logic_trace(log_n, seed=0, inputs=None):
    """orc_logic_trace: the AIR-2 witness [524, 2^log_n]; inputs [2^log_n, 9] (code, operand 0, operand 1) or seeded."""
    out = np.zeros((LOGIC_COLS, 1 << log_n), dtype=np.uint64)
    inp = np.ascontiguousarray(inputs, dtype=np.uint64) if inputs is not None else None
    if inp is not None:
        assert inp.shape == (1 << log_n, 9)
    lib().orc_logic_trace(seed, inp.ctypes.data if inp is not None else None, log_n, out)
    return out


def memory_trace(log_n, seed=0, inputs=None):
    """orc_memory_trace: the AIR-3 witness [45, 2^log_n] (the last column, the lookup's filter, zero); inputs [2^log_n, 11] (is_read, address, timestamp, value[8],
    sorted by address then timestamp) or a log drawn from the seed."""
    out = np.zeros((MEMORY_COLS, 1 << log_n), dtype=np.uint64)
    inp = np.ascontiguousarray(inputs, dtype=np.uint64) if inputs is not None else None
    if inp is not None:
        assert inp.shape == (1 << log_n, 11)
    lib().orc_memory_trace(seed, inp.ctypes.data if inp is not None else None, log_n, out)
    return out


def arithmetic_trace(log_n, seed=0, inputs=None):
    """orc_arithmetic_trace: the AIR-4 witness [309, 2^log_n]; inputs [2^log_n, 9] (code 0 none / 1 add / 2 sub / 3 lt /
    4 gt, x, y as four u64 each) or seeded."""
    out = np.zeros((ARITHMETIC_COLS, 1 << log_n), dtype=np.uint64)
    inp = np.ascontiguousarray(inputs, dtype=np.uint64) if inputs is not None else None
    if inp is not None:
        assert inp.shape == (1 << log_n, 9)
    lib().orc_arithmetic_trace(seed, inp.ctypes.data if inp is not None else None, log_n, out)
    return out


def byte_packing_trace(log_n, seed=0, inputs=None):
    """orc_byte_packing_trace: the AIR-5 witness [299, 2^log_n]; inputs [2^log_n, 6] (is_read, len, the 32 byte slots as
    four u64) or seeded."""
    out = np.zeros((BYTE_PACKING_COLS, 1 << log_n), dtype=np.uint64)
    inp = np.ascontiguousarray(inputs, dtype=np.uint64) if inputs is not None else None
    if inp is not None:
        assert inp.shape == (1 << log_n, 6)
    lib().orc_byte_packing_trace(seed, inp.ctypes.data if inp is not None else None, log_n, out)
    return out


def arithmetic_mul_trace(log_n, seed=0, inputs=None):
    """orc_arithmetic_mul_trace: the AIR-7 witness [1217, 2^log_n]; inputs [2^log_n, 9] (is_mul, x, y as four u64 each)
    or seeded."""
    out = np.zeros((ARITHMETIC_MUL_COLS, 1 << log_n), dtype=np.uint64)
    inp = np.ascontiguousarray(inputs, dtype=np.uint64) if inputs is not None else None
    if inp is not None:
        assert inp.shape == (1 << log_n, 9)
    lib().orc_arithmetic_mul_trace(seed, inp.ctypes.data if inp is not None else None, log_n, out)
    return out


def keccak_sponge_rows(msg: bytes):
    """orc_keccak_sponge_rows: -> (digest, [n_blocks, 44]) the sponge-table rows of one message"""
    n = lib().orc_keccak_sponge_rows(msg, len(msg), None, None)
    rows, digest = np.zeros((n, 44), dtype=np.uint64), C.create_string_buffer(32)
    lib().orc_keccak_sponge_rows(msg, len(msg), rows.ctypes.data, digest)
    return digest.raw, rows


def keccak_sponge_trace(log_n, seed=0, inputs=None):
    """orc_keccak_sponge_trace: the AIR-6 witness [2414, 2^log_n]; inputs [2^log_n, 44] (keccak_sponge_rows, flags 0 =
    padding row) or seeded."""
    out = np.zeros((KECCAK_SPONGE_COLS, 1 << log_n), dtype=np.uint64)
    inp = np.ascontiguousarray(inputs, dtype=np.uint64) if inputs is not None else None
    if inp is not None:
        assert inp.shape == (1 << log_n, 44)
    lib().orc_keccak_sponge_trace(seed, inp.ctypes.data if inp is not None else None, log_n, out)
    return out


class Committed:
    def __init__(self, handle, n_cols, log_n, rate_bits, cap_h):
        self.h, self.n_cols, self.log_n, self.rate_bits, self.cap_h = handle, n_cols, log_n, rate_bits, cap_h

    @classmethod
    def from_values(cls, values, rate_bits, cap_h, from_coeffs=False):
        values = arr(values)
        n_cols, n = values.shape
        f = lib().orc_commit_coeffs if from_coeffs else lib().orc_commit_values
        return cls(f(values, n.bit_length() - 1, n_cols, rate_bits, cap_h), n_cols, n.bit_length() - 1, rate_bits,
                   cap_h)

    def cap(self):
        p = lib().orc_committed_cap(self.h)
        return np.ctypeslib.as_array(p, shape=(1 << self.cap_h, 4)).copy()

    def lde(self):
        p = lib().orc_committed_lde(self.h)
        return np.ctypeslib.as_array(p, shape=(self.n_cols, 1 << (self.log_n + self.rate_bits))).copy()

    def coeffs(self):
        p = lib().orc_committed_coeffs(self.h)
        return np.ctypeslib.as_array(p, shape=(self.n_cols, 1 << self.log_n)).copy()

    def __del__(self):
        if self.h:
            lib().orc_committed_free(self.h)
            self.h = None


def synth_constants(seed, log_n, n_const):
    out = np.empty((n_const, 1 << log_n), dtype=np.uint64)
    lib().orc_synth_constants(seed, log_n, n_const, out)
    return out


def synth_trace(seed, cfg, consts=None):
    out = np.empty((cfg.n_cols, 1 << cfg.log_n), dtype=np.uint64)
    lib().orc_synth_trace(seed, C.byref(cfg), consts.ctypes.data if consts is not None else None, out)
    return out


def quotient_values(cfg, const_lde, trace_lde, aux_lde, ctl, alpha0, alpha1):
    """orc_quotient_values: natural-order LDE matrices [cols, n << rate_bits] -> [2, n << rate_bits]."""
    t, a = arr(trace_lde), arr(aux_lde)
    c = arr(const_lde) if const_lde is not None else None
    out = np.empty((2, t.shape[1]), dtype=np.uint64)
    lib().orc_quotient_values(C.byref(cfg), c.ctypes.data if c is not None else None, t, a,
                              np.ascontiguousarray(ctl, dtype=np.uint64), int(alpha0), int(alpha1), out)
    return out


def stark_prove(cfg, trace_values, ctl, challenger, consts_committed=None, trace_committed=None):
    """Runs the per-table prover.  The caller owns transcript setup (observe caps, draw ctl)."""
    tc = trace_committed or Committed.from_values(trace_values, cfg.rate_bits, cfg.cap_height)
    proof = np.zeros(lib().orc_proof_words(C.byref(cfg)), dtype=np.uint64)
    rc = lib().orc_stark_prove(C.byref(cfg), consts_committed.h if consts_committed else None, tc.h,
                               arr(trace_values), arr(ctl), C.byref(challenger.c), proof)
    if rc != 0:
        raise RuntimeError("orc_stark_prove failed: %d" % rc)
    return proof


def stark_verify(cfg, proof, ctl, challenger, const_cap=None):
    cc = arr(const_cap).ctypes.data if const_cap is not None else None
    return lib().orc_stark_verify(C.byref(cfg), cc, arr(ctl), C.byref(challenger.c), arr(proof))


def proof_digest(cfg, proof):
    out = np.empty(4, dtype=np.uint64)
    lib().orc_proof_digest(C.byref(cfg), arr(proof), out)
    return out


class PgState:
    """CPU restatement of ProverState + generate_{txn,agg,block}_proof (oracle/proofgen.c)."""

    def __init__(self, **kw):
        cfg = PgConfig()
        for k, v in kw.items():
            if k in ("table_log_lo", "table_log_hi"):
                for t in range(7):
                    getattr(cfg, k)[t] = v[t]
            else:
                setattr(cfg, k, v)
        self.cfg = cfg
        self.h = lib().orc_pg_state_build(C.byref(cfg))

    def _take(self, ptr, n):
        out = np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()
        lib().orc_free(ptr)
        return out

    def preprocess(self, ir_words):
        """Build the circuits a txn proof of this IR touches (lazy otherwise), so the proof can be timed alone."""
        rc = lib().orc_pg_preprocess(self.h, arr(ir_words))
        if rc:
            raise RuntimeError("orc_pg_preprocess failed: %d" % rc)

    WITNESS_WORDS = {0: 9, 1: 6, 3: 25, 4: 44, 5: 9, 6: 11}   # table index -> words per item (arithmetic, byte packing, keccak, logic, memory)

    def txn(self, ir_words, keccak_inputs=None, witness=None):
        """keccak_inputs: [n_perms, 25] permutation inputs of the txn's Keccak table (needs the IR's 0x100 flag).
        witness: {table index: [n, words] array} for any of the tables with an AIR (orc_pg_txn_witness)."""
        ptr, n = C.POINTER(C.c_uint64)(), C.c_size_t()
        if witness is not None:
            if keccak_inputs is not None:
                witness = {**witness, 3: keccak_inputs}
            keep, items, counts, given = [], (C.c_void_p * 7)(), (C.c_size_t * 7)(), (C.c_int * 7)()
            for t, data in witness.items():
                a = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1, self.WITNESS_WORDS[t])
                keep.append(a)
                items[t], counts[t], given[t] = (a.ctypes.data if a.size else None), a.shape[0], 1
            rc = lib().orc_pg_txn_witness(self.h, arr(ir_words), items, counts, given, C.byref(ptr), C.byref(n))
        elif keccak_inputs is None:
            rc = lib().orc_pg_txn(self.h, arr(ir_words), C.byref(ptr), C.byref(n))
        else:
            k = np.ascontiguousarray(keccak_inputs, dtype=np.uint64).reshape(-1, 25)
            rc = lib().orc_pg_txn_keccak(self.h, arr(ir_words), k.ctypes.data if k.size else None, k.shape[0], C.byref(ptr),
                                         C.byref(n))
        if rc:
            raise RuntimeError("orc_pg_txn failed: %d" % rc)
        return self._take(ptr, n)

    def txn_tables(self, ir_words, witness=None):
        """The seven table proofs of the transaction alone (upstream's AllProof; orc_pg_txn_tables), in the byte form of
        libbpg's bp_generate_txn_table_proofs.  witness as for txn()."""
        ptr, n = C.POINTER(C.c_uint64)(), C.c_size_t()
        keep, items, counts, given = [], (C.c_void_p * 7)(), (C.c_size_t * 7)(), (C.c_int * 7)()
        for t, data in (witness or {}).items():
            a = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1, self.WITNESS_WORDS[t])
            keep.append(a)
            items[t], counts[t], given[t] = (a.ctypes.data if a.size else None), a.shape[0], 1
        rc = lib().orc_pg_txn_tables(self.h, arr(ir_words), items, counts, given, C.byref(ptr), C.byref(n))
        if rc:
            raise RuntimeError("orc_pg_txn_tables failed: %d" % rc)
        return self._take(ptr, n)

    def verify_tables(self, table_proofs):
        """upstream's verify_proof on the table proofs: each table against the shared transcript, then the cross-table
        lookups (oracle/ctl.c).  0 = accept."""
        w = arr(table_proofs)
        return lib().orc_pg_verify_tables(C.byref(self.cfg), w, w.size)

    def agg(self, lhs, lhs_is_agg, rhs, rhs_is_agg):
        ptr, n = C.POINTER(C.c_uint64)(), C.c_size_t()
        lhs, rhs = arr(lhs), arr(rhs)
        rc = lib().orc_pg_agg(self.h, lhs, lhs.size, int(lhs_is_agg), rhs, rhs.size, int(rhs_is_agg), C.byref(ptr),
                              C.byref(n))
        if rc:
            raise RuntimeError("orc_pg_agg failed: %d" % rc)
        return self._take(ptr, n)

    def block(self, parent, agg):
        ptr, n = C.POINTER(C.c_uint64)(), C.c_size_t()
        agg = arr(agg)
        par = arr(parent) if parent is not None else None
        rc = lib().orc_pg_block(self.h, par.ctypes.data if par is not None else None, par.size if par is not None else 0,
                                agg, agg.size, C.byref(ptr), C.byref(n))
        if rc:
            raise RuntimeError("orc_pg_block failed: %d" % rc)
        return self._take(ptr, n)

    def verify(self, proof):
        proof = arr(proof)
        return lib().orc_pg_verify(self.h, proof, proof.size)

    def circuit_caps(self):
        """[3, 2^cap_height, 4]: root, agg, block circuit caps."""
        cw = 4 << self.cfg.stark_cap_height
        out = np.empty((3, cw), dtype=np.uint64)
        for k in range(3):
            lib().orc_pg_circuit_cap(self.h, k, out[k])
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_pg_state_free(self.h)
            self.h = None
