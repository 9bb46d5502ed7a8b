"""ctypes loader for libbpg.so (C ABI: include/bpg.h).  Fails loudly when the library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BPG_LIBBPG: another build of the same library (a host-sanitizer build, csrc/Makefile with OUT= / CXXFLAGS=)
_PATH = os.environ.get("BPG_LIBBPG") or os.path.join(_HERE, "lib", "libbpg.so")

BP_OK = 0
STATUS_NAMES = {0: "BP_OK", -1: "BP_ERR_ABORTED", -2: "BP_ERR_INVALID_INPUT", -3: "BP_ERR_RANGE",
                -4: "BP_ERR_DEVICE", -5: "BP_ERR_VERIFY", -6: "BP_ERR_UNSUPPORTED"}


class BpgError(RuntimeError):
    """Mirror of the reference's ProofGenError(String) (proof_gen.rs:22-36) with the status code."""

    def __init__(self, code, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(code, str(code)), message))
        self.code = code
        self.message = message


class StarkCfg(C.Structure):
    """bp_stark_cfg (include/bpg.h)."""
    _fields_ = [(n, C.c_uint32) for n in ("log_n", "n_cols", "n_const", "deg_pow", "rate_bits", "cap_height",
                                           "num_queries", "pow_bits", "arity_bits", "final_poly_bits")]


class AirFamily(C.Structure):
    """bp_air_family: a run of constraints of one kind and degree."""
    _fields_ = [(n, C.c_uint32) for n in ("first_index", "count", "kind", "degree")]


class AirDesc(C.Structure):
    """bp_air_desc (include/bpg.h)."""
    _fields_ = ([("air_id", C.c_uint32), ("name", C.c_char * 24)]
                + [(n, C.c_uint32) for n in ("fixed_n_cols", "n_const_max", "degree", "n_cols", "n_aux",
                                             "n_air_constraints", "n_ctl_constraints", "n_units", "n_families")]
                + [("families", AirFamily * 24)])


def take_buffer(ptr, length):
    """Copy a library-allocated buffer into bytes and release it with bp_free_buffer."""
    try:
        return C.string_at(ptr, length.value)
    finally:
        lib().bp_free_buffer(ptr)


def lib_path():
    return _PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_PATH):
        raise ImportError(
            "libbpg.so not built at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path." % _PATH)
    L = C.CDLL(_PATH)
    vp, u32, u64, i = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.bp_last_error.restype = C.c_char_p
    L.bp_version.restype = C.c_char_p
    L.bp_device_count.restype = i
    L.bp_ntt_batch.argtypes = [vp, u32, u32, u64, i, vp]
    L.bp_lde_batch.argtypes = [vp, u64, vp, u64, vp, u64, u32, u32, u32, i, vp]
    L.bp_poseidon_perm_batch.argtypes = [vp, u64, vp]
    L.bp_debug_field_ops.argtypes = [vp, vp, vp, u64, vp]
    L.bp_quotient_scratch_words.argtypes = [u32, C.POINTER(StarkCfg)]
    L.bp_quotient_scratch_words.restype = u64
    L.bp_quotient_eval.argtypes = [u32, C.POINTER(StarkCfg), vp, vp, vp, C.POINTER(u64), C.POINTER(u64), vp, vp, vp]
    L.bp_air_count.restype = u32
    L.bp_air_describe.argtypes = [u32, u32, u32, u32, C.POINTER(AirDesc)]
    L.bp_keccak_trace.argtypes = [vp, u64, u32, vp, vp]
    L.bp_logic_trace.argtypes = [vp, u64, u32, vp, vp]
    L.bp_memory_trace.argtypes = [vp, u64, u32, vp, vp]
    L.bp_arithmetic_trace.argtypes = [vp, u64, u32, vp, vp]
    L.bp_byte_packing_trace.argtypes = [vp, u64, u32, vp, vp]
    L.bp_keccak_sponge_trace.argtypes = [vp, u64, u32, vp, vp]
    L.bp_arithmetic_mul_trace.argtypes = [vp, u64, u32, vp, vp]
    L.bp_stark_verify_air.argtypes = [u32, C.POINTER(StarkCfg), C.POINTER(u64), C.c_char_p, C.c_size_t]
    L.bp_stark_prove_air.argtypes = [u32, C.POINTER(StarkCfg), u64, u64, i, C.POINTER(C.POINTER(C.c_uint8)),
                                     C.POINTER(C.c_size_t)]
    L.bp_fri_fold.argtypes = [vp, u32, u32, u32, u64, C.POINTER(u64), vp, vp]
    L.bp_openings.argtypes = [vp, u64, u32, u32, C.POINTER(u64), C.POINTER(u64), vp, vp, vp]
    L.bp_pow_grind.argtypes = [C.POINTER(u64), u32, u32, C.POINTER(u64), vp]
    L.bp_merkle_digest_words.argtypes = [u32, u32]
    L.bp_merkle_digest_words.restype = u64
    L.bp_merkle_commit.argtypes = [vp, u64, u32, u32, u32, u32, vp, vp]
    L.bp_stark_prove_synthetic.argtypes = [C.POINTER(StarkCfg), u64, u64, i, C.POINTER(C.POINTER(C.c_uint8)),
                                           C.POINTER(C.c_size_t)]
    L.bp_free_buffer.argtypes = [C.POINTER(C.c_uint8)]
    L.bp_free_buffer.restype = None
    L.bp_tune_quad_threshold.argtypes = [u64]
    L.bp_tune_quad_threshold.restype = None
    L.bp_tune_ntt_mx.argtypes = [i]
    L.bp_tune_ntt_mx.restype = None
    L.bp_tune_ntt_mx_wg_per_cu.argtypes = [i]
    L.bp_tune_ntt_mx_wg_per_cu.restype = None
    L.bp_tune_poseidon_mx.argtypes = [i]
    L.bp_tune_poseidon_mx.restype = None
    L.bp_tune_poseidon_mx_sets.argtypes = [i]
    L.bp_tune_poseidon_mx_sets.restype = None
    _lib = L
    return L


def check(rc):
    if rc != BP_OK:
        raise BpgError(rc, lib().bp_last_error().decode("utf-8", "replace"))
