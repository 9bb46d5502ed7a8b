"""L0 kernel-shaped operations on torch device buffers (thin wrappers over the C ABI).

torch is used only for device memory and streams (int64 tensors carry the u64 bit patterns).
Layouts are the ones documented in include/bpg.h: column-major matrices, natural-order values,
bit-reversed coefficients, coset-major LDE.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import StarkCfg, check, lib, take_buffer

NTT_FWD_BR2NAT, NTT_INV_NAT2BR, NTT_FWD_NAT, NTT_INV_NAT = 0, 1, 2, 3


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(t):
    if not t.is_cuda:
        raise ValueError("bpg ops need device tensors (there is no CPU fallback)")
    if t.dtype != torch.int64 or not t.is_contiguous():
        raise ValueError("bpg ops need contiguous int64 tensors holding u64 bit patterns")


def ntt_batch_(cols, direction):
    """In-place NTT of a [n_cols, n] column-major batch."""
    _require_cuda(cols)
    n_cols, n = cols.shape
    check(lib().bp_ntt_batch(cols.data_ptr(), n.bit_length() - 1, n_cols, n, direction, _stream()))
    return cols


def intt_batch(values, out=None):
    """Out-of-place inverse NTT: values [n_cols, n] natural -> coefficients (bit-reversed, scaled by 1/n)."""
    import ctypes as C
    _require_cuda(values)
    n_cols, n = values.shape
    out = torch.empty_like(values) if out is None else out
    L = lib()
    L.bp_intt_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    check(L.bp_intt_batch(values.data_ptr(), n, out.data_ptr(), n, n.bit_length() - 1, n_cols, _stream()))
    return out


def lde_batch(inp, rate_bits, from_coeffs=False):
    """values/coeffs [n_cols, n] -> (coeffs [n_cols, n] bit-reversed, lde [n_cols, n << rate_bits] coset-major)."""
    _require_cuda(inp)
    n_cols, n = inp.shape
    coeffs = torch.empty_like(inp)
    lde = torch.empty((n_cols, n << rate_bits), dtype=torch.int64, device=inp.device)
    check(lib().bp_lde_batch(inp.data_ptr(), n, coeffs.data_ptr(), n, lde.data_ptr(), n << rate_bits,
                             n.bit_length() - 1, rate_bits, n_cols, int(from_coeffs), _stream()))
    return coeffs, lde


def poseidon_perm_batch_(states):
    _require_cuda(states)
    check(lib().bp_poseidon_perm_batch(states.data_ptr(), states.numel() // 12, _stream()))
    return states


def field_ops(a, b):
    """bp_debug_field_ops: [15, n] planes of canonical results for operand vectors a, b (any u64 patterns)."""
    _require_cuda(a)
    _require_cuda(b)
    n = a.numel()
    out = torch.empty((15, n), dtype=torch.int64, device=a.device)
    check(lib().bp_debug_field_ops(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, _stream()))
    return out


AIR_SYNTHETIC, AIR_KECCAK_F = 0, 1
KECCAK_COLS = 2431
LOGIC_COLS = 524
MEMORY_COLS = 45
ARITHMETIC_COLS = 309
BYTE_PACKING_COLS = 299
KECCAK_SPONGE_COLS = 2414
ARITHMETIC_MUL_COLS = 1217


def air_describe(air_id, n_cols=0, n_const=0, deg_pow=1):
    """bp_air_describe: shape and constraint list of a built-in AIR."""
    from ._lib import AirDesc
    d = AirDesc()
    check(lib().bp_air_describe(air_id, n_cols, n_const, deg_pow, C.byref(d)))
    return d


def logic_trace(log_n, seed=0, inputs=None, device="cuda"):
    """bp_logic_trace: the AIR-2 witness [524, 2^log_n]; inputs [2^log_n, 9] int64 on the device (operation code, the
    four words of operand 0, of operand 1), or drawn from `seed`."""
    out = torch.empty((LOGIC_COLS, 1 << log_n), dtype=torch.int64, device=device)
    if inputs is not None:
        _require_cuda(inputs)
        assert inputs.shape == (1 << log_n, 9)
    check(lib().bp_logic_trace(inputs.data_ptr() if inputs is not None else None, seed, log_n, out.data_ptr(), _stream()))
    return out


def memory_trace(log_n, seed=0, inputs=None, device="cuda"):
    """bp_memory_trace: the AIR-3 witness [45, 2^log_n]; inputs [2^log_n, 11] int64 on the device (is_read, address,
    timestamp, eight value limbs; sorted by address then timestamp), or a log drawn from `seed`."""
    out = torch.empty((MEMORY_COLS, 1 << log_n), dtype=torch.int64, device=device)
    if inputs is not None:
        _require_cuda(inputs)
        assert inputs.shape == (1 << log_n, 11)
    check(lib().bp_memory_trace(inputs.data_ptr() if inputs is not None else None, seed, log_n, out.data_ptr(), _stream()))
    return out


def arithmetic_trace(log_n, seed=0, inputs=None, device="cuda"):
    """bp_arithmetic_trace: the AIR-4 witness [309, 2^log_n]; inputs [2^log_n, 9] int64 on the device (operation code
    0 none / 1 add / 2 sub / 3 lt / 4 gt, the four words of x, of y), or drawn from `seed`."""
    out = torch.empty((ARITHMETIC_COLS, 1 << log_n), dtype=torch.int64, device=device)
    if inputs is not None:
        _require_cuda(inputs)
        assert inputs.shape == (1 << log_n, 9)
    check(lib().bp_arithmetic_trace(inputs.data_ptr() if inputs is not None else None, seed, log_n, out.data_ptr(), _stream()))
    return out


def byte_packing_trace(log_n, seed=0, inputs=None, device="cuda"):
    """bp_byte_packing_trace: the AIR-5 witness [299, 2^log_n]; inputs [2^log_n, 6] int64 on the device (is_read, len,
    the 32 byte slots as four words), or drawn from `seed`."""
    out = torch.empty((BYTE_PACKING_COLS, 1 << log_n), dtype=torch.int64, device=device)
    if inputs is not None:
        _require_cuda(inputs)
        assert inputs.shape == (1 << log_n, 6)
    check(lib().bp_byte_packing_trace(inputs.data_ptr() if inputs is not None else None, seed, log_n, out.data_ptr(), _stream()))
    return out


def keccak_sponge_trace(log_n, seed=0, inputs=None, device="cuda"):
    """bp_keccak_sponge_trace: the AIR-6 witness [2414, 2^log_n]; inputs [2^log_n, 44] int64 on the device (flags, message
    bytes in the block, the block as absorbed, the state before it: proof_gen.keccak256_sponge_rows), or seeded."""
    out = torch.empty((KECCAK_SPONGE_COLS, 1 << log_n), dtype=torch.int64, device=device)
    if inputs is not None:
        _require_cuda(inputs)
        assert inputs.shape == (1 << log_n, 44)
    check(lib().bp_keccak_sponge_trace(inputs.data_ptr() if inputs is not None else None, seed, log_n, out.data_ptr(), _stream()))
    return out


def arithmetic_mul_trace(log_n, seed=0, inputs=None, device="cuda"):
    """bp_arithmetic_mul_trace: the AIR-7 witness [1217, 2^log_n]; inputs [2^log_n, 9] int64 on the device (is_mul, the
    four words of x, of y), or drawn from `seed`."""
    out = torch.empty((ARITHMETIC_MUL_COLS, 1 << log_n), dtype=torch.int64, device=device)
    if inputs is not None:
        _require_cuda(inputs)
        assert inputs.shape == (1 << log_n, 9)
    check(lib().bp_arithmetic_mul_trace(inputs.data_ptr() if inputs is not None else None, seed, log_n, out.data_ptr(), _stream()))
    return out


def keccak_trace(log_n, seed=0, inputs=None, device="cuda"):
    """bp_keccak_trace: the AIR-1 witness [2431, 2^log_n]; inputs [n_perm, 25] int64 lanes on the device, or drawn
    from `seed`."""
    out = torch.empty((KECCAK_COLS, 1 << log_n), dtype=torch.int64, device=device)
    if inputs is not None:
        _require_cuda(inputs)
        assert inputs.shape == (((1 << log_n) + 23) // 24, 25)
    check(lib().bp_keccak_trace(inputs.data_ptr() if inputs is not None else None, seed, log_n, out.data_ptr(), _stream()))
    return out


def quotient_eval(cfg, trace_lde, aux_lde, const_lde, ctl, alphas, air_id=AIR_SYNTHETIC):
    """bp_quotient_eval: [2, n << rate_bits] quotient values (coset-major) of AIR `air_id`."""
    _require_cuda(trace_lde)
    _require_cuda(aux_lde)
    if const_lde is not None:
        _require_cuda(const_lde)
    rows = trace_lde.shape[1]
    scratch = torch.empty(int(lib().bp_quotient_scratch_words(air_id, C.byref(cfg))), dtype=torch.int64,
                          device=trace_lde.device)
    out = torch.empty((2, rows), dtype=torch.int64, device=trace_lde.device)
    check(lib().bp_quotient_eval(air_id, C.byref(cfg), trace_lde.data_ptr(), aux_lde.data_ptr(),
                                 const_lde.data_ptr() if const_lde is not None else None,
                                 (C.c_uint64 * 4)(*[int(x) for x in ctl]), (C.c_uint64 * 2)(*[int(x) for x in alphas]),
                                 scratch.data_ptr(), out.data_ptr(), _stream()))
    return out


def fri_fold(values, log_nl, rate_bits, shift, beta, arity_bits=4):
    """bp_fri_fold: values [n_l << rate_bits, 2] (coset-major ext elements) -> next layer [(n_l >> 4) << rate_bits, 2]."""
    _require_cuda(values)
    out = torch.empty(((1 << (log_nl - arity_bits)) << rate_bits, 2), dtype=torch.int64, device=values.device)
    check(lib().bp_fri_fold(values.data_ptr(), log_nl, rate_bits, arity_bits, int(shift),
                            (C.c_uint64 * 2)(int(beta[0]), int(beta[1])), out.data_ptr(), _stream()))
    return out


def openings(coeffs, z0, z1=None):
    """bp_openings: coeffs [n_cols, n] bit-reversed -> [n_cols, 4] = (p(z0), p(z1)) as extension pairs."""
    _require_cuda(coeffs)
    n_cols, n = coeffs.shape
    pw = torch.empty(4 * n, dtype=torch.int64, device=coeffs.device)
    out = torch.zeros((n_cols, 4), dtype=torch.int64, device=coeffs.device)
    a0 = (C.c_uint64 * 2)(int(z0[0]), int(z0[1]))
    a1 = (C.c_uint64 * 2)(int(z1[0]), int(z1[1])) if z1 is not None else None
    check(lib().bp_openings(coeffs.data_ptr(), n, n.bit_length() - 1, n_cols, a0, a1, pw.data_ptr(), out.data_ptr(),
                            _stream()))
    return out


def pow_grind(state, pos, bits):
    """bp_pow_grind: smallest nonce for the 12-word sponge `state` (host ints) with `bits` leading zeros."""
    nonce = C.c_uint64()
    check(lib().bp_pow_grind((C.c_uint64 * 12)(*[int(x) for x in state]), pos, bits, C.byref(nonce), _stream()))
    return nonce.value


def merkle_commit(lde, log_n, rate_bits, cap_height):
    """Returns the level-order digest buffer [words/4, 4]; the last 2^cap_height rows are the cap."""
    _require_cuda(lde)
    n_cols, rows = lde.shape
    assert rows == 1 << (log_n + rate_bits)
    words = lib().bp_merkle_digest_words(log_n + rate_bits, cap_height)
    dig = torch.empty((words // 4, 4), dtype=torch.int64, device=lde.device)
    check(lib().bp_merkle_commit(lde.data_ptr(), rows, n_cols, log_n, rate_bits, cap_height, dig.data_ptr(),
                                 _stream()))
    return dig


def stark_cfg(log_n, n_cols, n_const=0, deg_pow=1, rate_bits=1, cap_height=4, num_queries=84, pow_bits=16,
              arity_bits=4, final_poly_bits=5):
    return StarkCfg(log_n, n_cols, n_const, deg_pow, rate_bits, cap_height, num_queries, pow_bits, arity_bits,
                    final_poly_bits)


def stark_prove_air(air_id, cfg, seed, const_seed=0, device=0):
    """One table proof on AIR `air_id`, witness generated on the device.  Returns proof words (u64)."""
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    check(lib().bp_stark_prove_air(air_id, C.byref(cfg), seed, const_seed, device, C.byref(out), C.byref(n)))
    return np.frombuffer(take_buffer(out, n), dtype=np.uint64).copy()


def stark_prove_synthetic(cfg, seed, const_seed=0, device=0):
    """One table proof on the synthetic AIR, witness generated on the device.  Returns proof words (u64)."""
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    check(lib().bp_stark_prove_synthetic(C.byref(cfg), seed, const_seed, device, C.byref(out), C.byref(n)))
    return np.frombuffer(take_buffer(out, n), dtype=np.uint64).copy()
