/* bpg.h -- C ABI of libbpg.so, the MI355X (gfx950) block-proof hot path.
 *
 * Drop-in boundary for the path BASELINE.json names: the three entry points of
 * plonky_block_proof_gen (reference: plonky_block_proof_gen/src/proof_gen.rs:39-43, :61-65,
 * :85-89), its prover/verifier state (prover_state.rs:17-20,80-100; verifier_state.rs:19-23,
 * 46-52,56-71) and, one level down, the kernel-shaped operations that a patched
 * PolynomialBatch::from_values / MerkleTree::new / compute_quotient_polys / fri_proof would call
 * (upstream plonky2 @ 265d46a9 -- not in the reference tree; the only in-tree anchors are the call
 * sites proof_gen.rs:44-52, :66-75, :97-103).  INTEGRATION.md shows the Rust `extern "C"` block a
 * maintainer would add.
 *
 * Conventions
 *  - plain C types only; device buffers are raw HIP device pointers (uint64_t*), streams are
 *    hipStream_t passed as void* (NULL = the null stream);
 *  - field elements are little-endian u64, canonical (< p = 2^64 - 2^32 + 1) on output; extension
 *    elements are two consecutive u64 (c0, c1), digests four;
 *  - every function returns 0 (BP_OK) or a negative bp_status; the message is in bp_last_error()
 *    (thread-local).  Nothing throws or aborts across the ABI (reference convention:
 *    Result<_, ProofGenError(String)>, proof_gen.rs:16-36);
 *  - inputs are caller-owned and read-only for the call; outputs returned through uint8_t** are
 *    library-allocated and released with bp_free_buffer (reference: results by value, children
 *    borrowed, proof_gen.rs:62-64,86-88);
 *  - a bp_state is immutable after bp_state_build and may be used from many threads at once
 *    (reference: &ProverState shared, proof_gen.rs:40).
 */
#ifndef BPG_H
#define BPG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  BP_OK = 0,
  BP_ERR_ABORTED = -1,       /* abort flag observed (proof_gen.rs:42,51) */
  BP_ERR_INVALID_INPUT = -2, /* malformed IR / proof bytes, non-contiguous children */
  BP_ERR_RANGE = -3,         /* trace taller/shorter than the configured table range (constants.rs:6-18) */
  BP_ERR_DEVICE = -4,        /* HIP error, no gfx950 device, out of device memory */
  BP_ERR_VERIFY = -5,        /* verifier rejected (verifier_state.rs:56-71) */
  BP_ERR_UNSUPPORTED = -6
} bp_status;

const char* bp_last_error(void);
const char* bp_version(void);
/* number of visible HIP devices; negative on error.  Does not initialise a context. */
int bp_device_count(void);

/* ------------------------------------------------------------------------------------------
 * L0 -- kernel-shaped operations on device buffers.
 *
 * Data layout in HBM: matrices are COLUMN-MAJOR, column c at base +
 * c*col_stride elements.  "values" are in natural row order.  "coeffs" are stored in
 * BIT-REVERSED order (position bitrev_n(j) holds coefficient j).  LDE evaluations are
 * "coset-major": position t*n + m holds the evaluation at 7 * w_{n*2^r}^(t + 2^r*m); the Merkle
 * leaf index of that row is bitrev_r(t)*n + bitrev_n(m), i.e. exactly upstream's
 * reverse_index_bits order.
 * ------------------------------------------------------------------------------------------ */

/* K2.  plonky2_field fft/ifft semantics on a batch of columns, in place.
 *   dir = BP_NTT_FWD_BR2NAT: coefficients (bit-reversed order) -> values (natural)
 *   dir = BP_NTT_INV_NAT2BR: values (natural) -> coefficients (bit-reversed), includes 1/n
 *   dir = BP_NTT_FWD_NAT / BP_NTT_INV_NAT: natural in, natural out (adds one permutation pass) */
enum { BP_NTT_FWD_BR2NAT = 0, BP_NTT_INV_NAT2BR = 1, BP_NTT_FWD_NAT = 2, BP_NTT_INV_NAT = 3 };
int bp_ntt_batch(uint64_t* d_cols, uint32_t log_n, uint32_t n_cols, uint64_t col_stride, int dir,
                 void* stream);

/* K2.  PolynomialBatch::from_values / from_coeffs low-degree extension.
 *   d_in: n_cols columns of n values (natural) or, if from_coeffs, n coefficients (bit-reversed);
 *   d_coeffs_out (nullable unless !from_coeffs... may alias nothing): n_cols x n coefficients, bit-reversed;
 *   d_lde_out: n_cols x (n << rate_bits), coset-major.  Strides in elements. */
int bp_lde_batch(const uint64_t* d_in, uint64_t in_stride, uint64_t* d_coeffs_out, uint64_t coeffs_stride,
                 uint64_t* d_lde_out, uint64_t lde_stride, uint32_t log_n, uint32_t rate_bits,
                 uint32_t n_cols, int from_coeffs, void* stream);

/* K3.  Poseidon-Goldilocks permutation (width 12) on n states of 12 words, in place. */
int bp_poseidon_perm_batch(uint64_t* d_states, uint64_t n, void* stream);

/* K4.  MerkleTree::new(leaves, cap_height) over the rows of a coset-major LDE matrix.
 *   n_leaves = n << rate_bits rows of n_cols elements; leaf digest = hash_or_noop(row).
 *   d_digests: level-order buffer of bp_merkle_digest_words(log_leaves, cap_height) words
 *   (leaf digests in leaf-index order first, ..., the 2^cap_height cap digests last). */
uint64_t bp_merkle_digest_words(uint32_t log_leaves, uint32_t cap_height);
int bp_merkle_commit(const uint64_t* d_lde, uint64_t lde_stride, uint32_t n_cols, uint32_t log_n,
                     uint32_t rate_bits, uint32_t cap_height, uint64_t* d_digests, void* stream);

/* ------------------------------------------------------------------------------------------
 * L0.5 -- one table proof on the synthetic AIR: K2-K9 end to end.
 * This is what plonky2_evm's prove_single_table does for one STARK table (reached from
 * proof_gen.rs:44-52); it exists in the ABI so the whole per-table path can be parity-tested
 * against the oracle.  Transcript prologue: observe constants cap (if n_const), observe trace cap,
 * draw 4 CTL challenges.  The witness is generated on the device from (seed, const_seed).
 * *out receives orc-identical proof words (little-endian u64), *out_len their byte length.
 * ------------------------------------------------------------------------------------------ */
typedef struct bp_stark_cfg {
  uint32_t log_n, n_cols, n_const, deg_pow, rate_bits, cap_height, num_queries, pow_bits, arity_bits,
      final_poly_bits;
} bp_stark_cfg;
int bp_stark_prove_synthetic(const bp_stark_cfg* cfg, uint64_t seed, uint64_t const_seed, int device,
                             uint8_t** out, size_t* out_len);
void bp_free_buffer(uint8_t* buf);

#ifdef __cplusplus
}
#endif
#endif /* BPG_H */
