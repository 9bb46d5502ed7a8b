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
/* Makes host threads SLEEP while they wait for this device (hipDeviceScheduleBlockingSync) instead of
 * spinning.  A prover stream per host thread needs this to use more streams than the host has cores.
 * The mode is decided ONCE per device, by whoever comes first -- this call, or the first worker the
 * library creates on the device (bp_state_build, bp_stark_prove_air) -- and is never changed
 * afterwards: later calls return BP_OK and do nothing.  The mode can only be switched on a device the
 * PROCESS has not used yet (switching it under queues that already carried work makes a later hipFree or
 * hipDeviceSynchronize hang on this ROCm): on a device that is already in use the call leaves the mode
 * alone, and the library's prover threads then wait with a loop of their own -- poll the event, sleep an
 * eighth of the time waited so far between polls -- instead of the runtime's spinning wait (measured on
 * 16 streams: within 1 % of the interrupt-driven rate at 10 % of a core per thread; the runtime's wait
 * keeps every thread at 100 %).  A process that uses the device through another library (PyTorch, ...)
 * may still call this first thing to get interrupt-driven waits. */
int bp_use_blocking_sync(int device);
/* What was decided for the device: 0 nothing yet, 1 host waits sleep, 2 the device was already in use (or
 * configured otherwise) and keeps its mode. */
int bp_host_wait_mode(int device);

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

/* K2.  The inverse transform out of place (what PolynomialBatch::from_values does first: the values stay, the
 * coefficients, bit-reversed and scaled by 1/n, go to d_coeffs_out).  The two buffers must be the same pointer
 * (in place) or not overlap. */
int bp_intt_batch(const uint64_t* d_values, uint64_t in_stride, uint64_t* d_coeffs_out, uint64_t out_stride,
                  uint32_t log_n, uint32_t n_cols, void* stream);

/* K2.  PolynomialBatch::from_values / from_coeffs low-degree extension.
 *   d_in: n_cols columns of n values (natural) or, if from_coeffs, n coefficients (bit-reversed);
 *   d_coeffs_out (nullable if from_coeffs): n_cols x n coefficients, bit-reversed; the same pointer as d_in
 *   (in place) or not overlapping it;
 *   d_lde_out: n_cols x (n << rate_bits), coset-major, overlapping neither.  Strides in elements.
 * Inputs may be any u64 (reduced mod p on the way in), outputs are canonical; overlapping buffers are refused
 * with BP_ERR_INVALID_INPUT. */
int bp_lde_batch(const uint64_t* d_in, uint64_t in_stride, uint64_t* d_coeffs_out, uint64_t coeffs_stride,
                 uint64_t* d_lde_out, uint64_t lde_stride, uint32_t log_n, uint32_t rate_bits,
                 uint32_t n_cols, int from_coeffs, void* stream);

/* Measurement aid: plain 8-byte-per-lane streaming copy of n words (calibrates PMC byte counters). */
int bp_debug_copy_u64(const uint64_t* d_in, uint64_t* d_out, uint64_t n, void* stream);

/* K1 self-check: every device form of the Goldilocks arithmetic (types.rs:10 fixes the field) on n
 * operand pairs, for tests against big-integer arithmetic.  d_a/d_b: any u64 values (non-canonical
 * allowed); d_out: 15 planes of n canonical words: a*b by the one-element carry chain, in groups of
 * four, in groups of three, by the compiler form; a+b; a-b; the unreduced dot-product accumulator
 * (2*a*b + a*b[first of its group of four]); 7*a; a^-1; the two components of the extension product
 * (a, b) * (b, a^b); then the interleaved group forms the NTT butterflies use: a + canon(b), a - canon(b),
 * canon(a), canon(b). */
int bp_debug_field_ops(const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, uint64_t n, void* stream);

/* Tuning knob: the size (rows / nodes / candidates of one launch) from which the hashing kernels put 64 Poseidon
 * states on a wave.  Matrix-core form (default): four sets of 16 states per wave from this size up, two from half of
 * it, one below.  With bp_tune_poseidon_mx(0): one lane per state from this size up, the quad-cooperative kernels
 * (4 lanes per state, DPP exchange) below.  Results are identical either way.  0 (default) = automatic: 2^19 (2^17
 * without the matrix-core form) while fewer than 6 provers (bp_generate_*_proof calls) are at work on the device,
 * 2^13 under load.  A caller that drives the L0 entry points from many streams itself should set 2^13: the library
 * cannot see that load. */
void bp_tune_quad_threshold(uint64_t n_perms);
/* Merkle levels of at most 2048 nodes: 1 (default) = fused, up to 7 levels per launch (LDS hand-down, one-set
 * matrix-core permutation; +2.8 % on the 256-txn block, +3..5 % on 16- and 32-txn shards: fewer launches on every
 * proof's critical path); 0 = one launch per level; -1 = fused only while fewer than 6 provers are at work on the
 * device.  Results are identical. */
void bp_tune_merkle_fused(int mode);
/* With the fused tail on: levels of up to 2^k parents (k = 8..24; 0 = none) also go nine at a time through
 * merkle_subtree_wide_kernel (256 parents per workgroup, four-set waves while a level has 64 or more of them).
 * Results are identical. */
void bp_tune_merkle_wide(int log2_parents);
/* 1 (default): hashing launches at or above the quad threshold use the matrix-core form of the permutation
 * (csrc/poseidon_mx.cuh: the MDS layer as int8 MFMAs on the byte planes of the state); 0: one lane per state.
 * Results are identical either way. */
void bp_tune_poseidon_mx(int on);
/* Sets of 16 states a wave of the matrix-core form carries: 4, 2 or 1 (fewer sets = more waves for the same launch);
 * 0 (default) = by launch size: 4 from the quad threshold up, 2 from half of it, else 1.  Results are identical. */
void bp_tune_poseidon_mx_sets(int sets);
/* The four-set matrix-core kernels take the partial rounds in groups: within a group only the next S-box input (an
 * affine form of the untouched words and the earlier S-box outputs, evaluated by int8 MFMAs on their bytes) is
 * recombined per round, the twelve words once per group (csrc/poseidon_mx.cuh, grp; tools/poseidon_group_model.py).
 * 3 (default; any other non-zero value means 3): all 22 partial rounds, as 8 + 8 + 6; 2: rounds 4..19 as 8 + 8, rounds
 * 20..25 one by one; 0: every round by itself.  Results are identical in every mode. */
void bp_tune_poseidon_grouped(int mode);
/* The load-dependent choices -- Poseidon sets per wave by launch size, K5 and the FRI alpha-combination in one pass
 * without partial sums -- follow the number of bp_generate_*_proof calls at work on the device (six or more =
 * loaded): -1 (default).  0 / 1: stated by a caller that drives the L0 / L0.5 entry points from its own threads (the
 * library cannot see that load), or by a test that pins both paths.  Results are identical. */
void bp_tune_assume_loaded(int mode);
/* The recursion-shaped proofs of a transaction -- level k of its seven per-table chains (proof_gen.rs:44-52 proves all
 * tables in one call; upstream shrinks each table's proof through a chain of fixed-shape circuits) -- are proved in
 * lock-step, up to n (1..8, default 8) proofs per batch: seven transcripts stepped together, every kernel launch and every
 * host wait shared.  1 = one proof at a time (round 3's behaviour, for A/B runs).  Results are identical. */
void bp_tune_rec_batch(int n);
/* A prover that is alone on the device (a lone transaction, the last one of a shard) spreads its transaction's seven
 * trace commitments -- which do not depend on each other -- over the streams of up to three idle workers of the state
 * (1, default); 0 = always on the prover's own stream; n > 1 = also while up to n provers are at work (measured: no gain,
 * profiles/r5_block_size_series.txt).  Results are identical. */
void bp_tune_side_lanes(int n);
/* The witness of a recursion circuit's Poseidon rows (the sponge over its public-input list, its children's Merkle paths,
 * the sponges over their opened rows: independent pieces) is made on the host; a prover that is alone on the device makes
 * the pieces of a lock-step batch on up to n threads (default 7; 1 = on the prover's own thread).  Results are identical. */
void bp_tune_witness_threads(int n);
/* How the library's prover threads wait for the device: 0 (default) = the runtime's wait where it sleeps
 * (bp_host_wait_mode 1), the library's own poll-and-sleep wait where the runtime's would spin (mode 2: a device the
 * process had already used when the library came to it); 1 = always the runtime's wait; 2 = always poll and sleep. */
void bp_tune_host_wait(int mode);
/* The CPU permutation of the Fiat-Shamir transcript (K7 stays on the host): 0 (default) = the AVX2 form of the MDS
 * layer where the CPU has it, 1 = always the scalar form.  Same values either way (tests). */
void bp_tune_host_poseidon(int mode);
/* Host only: that permutation over n states of 12 words (any u64 in, canonical out), in place. */
int bp_debug_poseidon_host(uint64_t* states, size_t n);
/* Measurement knob: 1 = while the device is loaded the quotient kernel spreads the units of the SYNTHETIC AIR over
 * workgroup rows until the launch has 256 workgroups, as it does for the AIRs of the real tables; 0 (default): one pass. */
void bp_tune_k5_spread(int on);
/* Host only: the operand images of one group of K partial rounds starting at round r0 as the device gets them
 * (csrc/poseidon_group.hpp); tests/test_mx_tables.py pins them to tools/poseidon_group_model.py.
 * out_ops: bp_debug_poseidon_group_ops(K) x 1024 bytes, out_cform: 64 i32, out_cmain: 96 i32. */
uint32_t bp_debug_poseidon_group_ops(uint32_t K);
int bp_debug_poseidon_group_tables(uint32_t K, uint32_t r0, uint8_t* out_ops, int32_t* out_cform, int32_t* out_cmain,
                                   int32_t* out_max_plane_sum);

/* Tuning knob for K2: 0 (default) = automatic, 1 = never, 2 = wherever possible: transform a 2^13 / 2^14-point
 * block with TWO workgroups that each do half of the stage coupling its halves while loading (csrc/ntt.hip).
 * Results are identical either way. */
void bp_tune_ntt_split(int mode);
/* 2^14-point coset-LDE blocks as persistent workgroups that prefetch the next block's coefficients while the last pass
 * of the current one computes and stores (csrc/ntt.hip, ntt16_dit_persist_kernel): on = 1 / 0 = the one-shot grid (the
 * default: the persistent form measured 8 % slower, profiles/r5_ntt_stalls.txt); resident_workgroups > 0 sets the grid (default 256 = one per CU).  Same values either way (tests). */
void bp_tune_ntt_persist(int on, int resident_workgroups);
/* NTT blocks as three radix-16 passes whose 16-point DFTs are int8 MFMAs on the bytes of the elements
 * (csrc/ntt_mx.cuh): 0 = never (the VALU butterfly kernels everywhere), 1 = 2^12- and 2^13-point blocks, 2 = 2^14-point
 * blocks too, 3 (default) = 2^13-point blocks while fewer than 6 provers are at work on the device, 4 / 5 = like 1 for
 * the inverse (DIF) / forward (DIT) direction only (measurement).  Alone
 * on the chip the form is level to +17 %; under the multi-stream block run its register footprint loses 12 %
 * (DESIGN.md section 7).  Results are identical either way. */
void bp_tune_ntt_mx(int mode);
/* Measurement knob: resident workgroups per CU of the (persistent) matrix-core NTT kernels; 0 = default. */
void bp_tune_ntt_mx_wg_per_cu(int n);
/* Host only (no GPU needed): the constants the matrix-core kernels run on, as the device gets them, so that CPU tests
 * can pin them to an independent derivation.  NTT (csrc/ntt_mx.cuh): kind 0 = DIF / 1 = DIT matrix, inverse = root
 * direction; out_a 8192 bytes ([row block 8][lane 64][16] int8, K-chunk 0), out_c 128 i32, out_tw256 4096 u64,
 * out_tw16 256 u64.  Poseidon (csrc/poseidon_mx.cuh): the C-operand table, 30 x 4 x 24 u32. */
int bp_debug_ntt_mx_tables(int kind, int inverse, uint8_t* out_a, int32_t* out_c, uint64_t* out_tw256,
                           uint64_t* out_tw16);
int bp_debug_poseidon_mx_cin(uint32_t* out);

/* K3.  Poseidon-Goldilocks permutation (width 12) on n states of 12 words, in place. */
int bp_poseidon_perm_batch(uint64_t* d_states, uint64_t n, void* stream);

/* K4.  MerkleTree::new(leaves, cap_height) over the rows of a coset-major LDE matrix.
 *   n_leaves = n << rate_bits rows of n_cols elements; leaf digest = hash_or_noop(row).
 *   d_digests: level-order buffer of bp_merkle_digest_words(log_leaves, cap_height) words
 *   (leaf digests in leaf-index order first, ..., the 2^cap_height cap digests last). */
uint64_t bp_merkle_digest_words(uint32_t log_leaves, uint32_t cap_height);
int bp_merkle_commit(const uint64_t* d_lde, uint64_t lde_stride, uint32_t n_cols, uint32_t log_n,
                     uint32_t rate_bits, uint32_t cap_height, uint64_t* d_digests, void* stream);

struct bp_stark_cfg;
/* K5.  The AIRs the library can prove (csrc/air.hpp).  What upstream expresses as `impl Stark for ...`
 * (eval_packed_generic / eval_ext, evaluated by plonky2_evm's compute_quotient_polys, reached from
 * proof_gen.rs:44-52; the seven zkEVM tables of prover_state.rs:85-93, Keccak range constants.rs:12) is here an
 * air_id: 0 = the synthetic AIR of DESIGN.md section 4 (any width), 1 = keccak_f, one round of Keccak-f[1600] per
 * row on 2431 columns, written from FIPS 202 (not upstream's column layout), 2 = logic, one AND / OR / XOR of two
 * 256-bit words per row on 524 columns, 3 = memory, a log of reads and writes sorted by (address, timestamp) on 45
 * columns, 4 = arithmetic, ADD / SUB / LT / GT on 256-bit words with a carry chain on 309 columns, 5 = byte_packing,
 * a big-endian byte sequence and the word it spells on 299 columns, 6 = keccak_sponge, the absorbing side of
 * Keccak-256 (XOR into the rate, chaining, pad10*1) on 2414 columns, 7 = arithmetic_mul, x * y = z + 2^256 w on 1217
 * columns (likewise their own layouts; a transaction's arithmetic table is proven by AIR 4 or by AIR 7: bp_ir_set_arithmetic_mul_air), 8 = plonk, a
 * PLONK-shaped circuit as a table: 135 wires (80 routed), 84 preprocessed constant columns (two gate selectors, two gate
 * constants, 80 sigmas), arithmetic and S-box gates, public inputs bound to the first row and the copy-constraint
 * permutation argument (Z + nine partial products per challenge set) as its auxiliary columns; degree 9, rate_bits 3 -- the
 * proof system of upstream's recursion circuits (CircuitData::prove), not their gate set and not a verifier circuit.  bp_air_describe returns the shape and
 * the constraint list of an AIR as families (first index, count, kind, degree); the list is followed by the constraints
 * of the table's auxiliary columns, its cross-table lookups (csrc/air.hpp, namespace ctl): n_cols / 8 unfiltered running
 * products for the synthetic AIR (a load placeholder), filter + carried input + two filtered running products for
 * keccak_f (the looked side of keccak_sponge -> keccak_f), two filtered running products for keccak_sponge (the looking
 * side), two for byte_packing (the looking side of byte_packing -> memory), a filter and two for memory (the looked
 * side), one constant product for the tables no lookup is built for. */
typedef struct bp_air_family {
  uint32_t first_index, count;
  uint32_t kind;   /* 0 all rows, 1 transition (x (X - g^(n-1))), 2 first row (x L_0), 3 last row (x L_(n-1)) */
  uint32_t degree; /* in the trace polynomials, before the selector */
} bp_air_family;
typedef struct bp_air_desc {
  uint32_t air_id;
  char name[24];
  uint32_t fixed_n_cols;      /* 0: the AIR takes any width */
  uint32_t n_const_max;       /* preprocessed constant columns it can use */
  uint32_t degree;            /* constraint degree: rate_bits must give 2^rate_bits >= degree - 1 */
  uint32_t n_cols, n_aux;     /* for the width asked about */
  uint32_t n_air_constraints, n_ctl_constraints;
  uint32_t n_units;           /* independently evaluable slices of the list (the kernel's grid.y granularity) */
  uint32_t n_families;
  bp_air_family families[24]; /* interleaved families (synthetic AIR, CTL) list the index of their first member */
} bp_air_desc;
uint32_t bp_air_count(void);
int bp_air_describe(uint32_t air_id, uint32_t n_cols, uint32_t n_const, uint32_t deg_pow, bp_air_desc* out);

/* K5.  Constraint / quotient evaluation on the extended domain: what compute_quotient_polys does for one table.
 * shape: log_n, n_cols, n_const, deg_pow, rate_bits of the table (the other fields must make a valid
 * configuration: use the values of the proof the quotient belongs to).  The three LDE matrices are
 * column-major and coset-major with column stride n << rate_bits (d_aux_lde: bp_air_desc.n_aux columns; d_const_lde may
 * be NULL when n_const == 0).  ctl = beta0, gamma0, beta1, gamma1; alphas = the two constraint challenges.
 * d_scratch: bp_quotient_scratch_words(air_id, shape) words.  d_qvals_out: [2][n << rate_bits], coset-major:
 * position t*n + m = quotient value at 7 * w_{n 2^r}^(t + 2^r m), already divided by Z_H.
 * One kernel serves every AIR; the random linear combination stays in registers, and a table tall enough to fill
 * the chip is evaluated in one pass without partial sums in HBM. */
uint64_t bp_quotient_scratch_words(uint32_t air_id, const struct bp_stark_cfg* shape);
int bp_quotient_eval(uint32_t air_id, const struct bp_stark_cfg* shape, const uint64_t* d_trace_lde,
                     const uint64_t* d_aux_lde, const uint64_t* d_const_lde, const uint64_t ctl[4],
                     const uint64_t alphas[2], uint64_t* d_scratch, uint64_t* d_qvals_out, void* stream);

/* Witness of AIR 1 (generate_traces is inside the reference's call too, proof_gen.rs:44-52): n = 2^log_n rows x 2431
 * columns (the last one, the lookup's filter, zero), column-major, row r = round r % 24 of permutation r / 24.  d_inputs: [ceil(n / 24)][25] input lanes
 * (any u64; lane x + 5y), or NULL to draw them from `seed` (splitmix64(seed ^ (lane << 32) ^ permutation)). */
int bp_keccak_trace(const uint64_t* d_inputs, uint64_t seed, uint32_t log_n, uint64_t* d_trace_out, void* stream);
/* Witness of AIR 2 (the logic table: one AND / OR / XOR of two 256-bit words per row): n = 2^log_n rows x 524 columns
 * (the last one, the filter of the lookup keccak_sponge -> logic, zero), column-major.  d_inputs: [n][9] = operation code (0 none = a padding row, 1 and, 2 or, 3 xor), then the four 64-bit
 * words of operand 0 and of operand 1, least significant first; or NULL to draw them from `seed`
 * (code = splitmix64(seed ^ (0xFF << 32) ^ row) & 3, word w of operand j = splitmix64(seed ^ ((1 + 4 j + w) << 32) ^ row)). */
int bp_logic_trace(const uint64_t* d_inputs, uint64_t seed, uint32_t log_n, uint64_t* d_trace_out, void* stream);
/* Witness of AIR 3 (the memory table: a log of reads and writes sorted by address, then timestamp): n = 2^log_n rows x
 * 45 columns (the last one, the lookup's filter, zero), column-major.  d_inputs: [n][11] = is_read, address (< 2^32), timestamp (< 2^32), eight 32-bit value
 * limbs, ALREADY SORTED (the kernel derives the address_changed flag and the gap bits from neighbouring rows; a log
 * that is out of order, or whose reads do not return the previous value, gives a witness the verifier rejects); or NULL
 * for a log drawn from `seed` (four operations per address; csrc/stark_kernels.hip, memory_trace_kernel). */
int bp_memory_trace(const uint64_t* d_inputs, uint64_t seed, uint32_t log_n, uint64_t* d_trace_out, void* stream);
/* Witness of AIR 4 (the additive part of the arithmetic table: ADD / SUB / LT / GT on 256-bit words as sixteen 16-bit
 * limbs with a carry chain): n = 2^log_n rows x 309 columns, column-major.  d_inputs: [n][9] = operation code (0 none,
 * 1 add, 2 sub, 3 lt, 4 gt), then the four 64-bit words of x and of y, least significant first; or NULL to draw them
 * from `seed` (code = splitmix64(seed ^ (0xFE << 32) ^ row) % 5, words as in bp_logic_trace). */
int bp_arithmetic_trace(const uint64_t* d_inputs, uint64_t seed, uint32_t log_n, uint64_t* d_trace_out, void* stream);
/* Witness of AIR 5 (the byte-packing table: a big-endian sequence of 1..32 bytes and the 256-bit word it spells, what
 * MLOAD_32BYTES / MSTORE_32BYTES move): n = 2^log_n rows x 299 columns, column-major.  d_inputs: [n][6] = word 0:
 * is_read (bit 0) | timestamp << 8; word 1: len (low byte; 0 = a padding row; above 32: 32) | address << 8 -- the 32-bit
 * address and timestamp of the memory operation that moves the word (the lookup byte_packing -> memory sends
 * (is_read, address, timestamp, value limbs) to the memory table; zero when the caller does not care) --; then the 32
 * byte slots as four 64-bit words (slot i = byte i % 8 of word i / 8; slots from len on are ignored); or NULL to draw
 * them from `seed` (address = row, timestamp = 2 + row). */
int bp_byte_packing_trace(const uint64_t* d_inputs, uint64_t seed, uint32_t log_n, uint64_t* d_trace_out, void* stream);
/* Witness of AIR 6 (the Keccak sponge table: the absorbing side of Keccak-256, one 136-byte block per row): n =
 * 2^log_n rows x 2414 columns, column-major.  d_inputs: [n][44] = flags (1 full block, 2 final block, 0 padding row),
 * message bytes in the block, the block as absorbed (17 words, pad10*1 included), the 25 lanes of the state before the
 * block -- bp_keccak256_sponge_rows makes them for a message; or NULL for one single-block message per row drawn from
 * `seed`.  The kernel computes the XOR and the permutation of every row. */
int bp_keccak_sponge_trace(const uint64_t* d_inputs, uint64_t seed, uint32_t log_n, uint64_t* d_trace_out, void* stream);
/* Witness of AIR 7 (the multiplicative half of the arithmetic table, a table of its own here: x * y = z + 2^256 w as a
 * 32-column schoolbook product over 16-bit limbs with 21-bit carries): n = 2^log_n rows x 1217 columns, column-major.
 * d_inputs: [n][9] = is_mul (0 = a padding row), the four 64-bit words of x and of y; or NULL to draw them from `seed`. */
int bp_arithmetic_mul_trace(const uint64_t* d_inputs, uint64_t seed, uint32_t log_n, uint64_t* d_trace_out, void* stream);

/* AIR 8 (plonk): the 85 preprocessed constant columns of the fixed circuit (selectors, gate constants drawn from `seed`,
 * the Poseidon-row selector, the 80 sigmas of its copy permutation), n = 2^log_n rows, column-major, for a circuit that
 *   - hashes a public-input list of pi_len words (1..104) in Poseidon-gate rows 4.. (one permutation per row), and
 *   - walks n_paths Merkle paths of path_depth levels each in Poseidon-gate rows 17.. (n_paths x path_depth <= 96): path p
 *     starts at list words path_pi0 + 8p .. + 3 (a leaf digest) and must arrive at list words path_pi0 + 8p + 4 .. + 7 (a
 *     cap entry); every recursion circuit of the prover state walks one path per child proof (merkle_proofs::
 *     verify_merkle_proof_to_cap of the child's first query into its trace oracle).  Arithmetic rows start at
 *     the first multiple of four past the Merkle rows (20 without paths) and one group of four must fit (2^log_n >= 32);
 * and the circuit's witness, 135 wires: free wires drawn from `seed`, the list pi hashed in rows 4.., its hash in row 0
 * (the four public inputs) and, through copy constraints, in the first arithmetic row; `paths` (NULL when n_paths = 0):
 * per path 1 + 4 path_depth + leaf_len words -- the leaf's position (bit l = "the node of level l is a right child"), the
 * sibling digests from the leaf upward, then (leaf_len > 0: the aggregation and block circuits of a prover state whose
 * recursion shape has the rows for it) the leaf_len words of the opened row, which the circuit hashes in ceil(leaf_len / 8)
 * Poseidon rows per path right after the Merkle rows; their digest is the path's first node and the list's leaf digest.  A witness whose path does not arrive at the list's cap entry is written as it
 * is: its proof is what a verifier rejects.  bp_plonk_trace returns after the stream has run it. */
typedef struct bp_plonk_layout {
  uint32_t pi_len, n_paths, path_depth, path_pi0;
  uint32_t leaf_len; /* 0, or the words of the row each path's leaf digest is the hash of (> 8): the circuit then hashes it too */
} bp_plonk_layout;
int bp_plonk_constants(uint64_t seed, uint32_t log_n, const bp_plonk_layout* layout, uint64_t* d_consts_out, void* stream);
int bp_plonk_trace(const uint64_t* d_consts, uint64_t seed, const uint64_t* pi, const bp_plonk_layout* layout, const uint64_t* paths,
                   uint32_t log_n, uint64_t* d_trace_out, void* stream);

/* K6.  One FRI fold (plonky2 fri::prover::fri_committed_trees: reduce_with_powers(beta) + coset_fft on the
 * folded domain), done in the evaluation domain.  d_values: the layer's n_l << rate_bits extension values
 * (c0, c1 interleaved) on shift * <w_{n_l 2^r}>, coset-major (position t*n_l + m = point index t + 2^r m).
 * d_out: the next layer, (n_l / 2^arity_bits) << rate_bits values on shift^(2^arity_bits) * <...>, same
 * layout.  arity_bits must be 4 (ConstantArityBits(4, 5)). */
int bp_fri_fold(const uint64_t* d_values, uint32_t log_nl, uint32_t rate_bits, uint32_t arity_bits, uint64_t shift,
                const uint64_t beta[2], uint64_t* d_out, void* stream);

/* K8.  Openings (plonky2_evm StarkOpeningSet::new / PolynomialBatch eval): every column polynomial of a
 * bit-reversed coefficient matrix (as bp_lde_batch leaves it; column stride in words) evaluated at one
 * or two extension-field points z0, z1 (z1 may be NULL).  d_pw_scratch: 4 << log_n words.
 * d_out: 4 words per column = (p(z0).c0, p(z0).c1, p(z1).c0, p(z1).c1); the z1 pair is zero when z1 is NULL. */
int bp_openings(const uint64_t* d_coeffs, uint64_t stride, uint32_t log_n, uint32_t n_cols, const uint64_t z0[2],
                const uint64_t z1[2], uint64_t* d_pw_scratch, uint64_t* d_out, void* stream);

/* K9.  Proof-of-work grind (plonky2 fri::prover::fri_proof_of_work): the SMALLEST nonce w such that the
 * Poseidon permutation of `state` with word `pos` (a rate word, < 8) replaced by w has `bits` leading
 * zero bits in output word 7.  Upstream accepts any witness; the minimum makes results reproducible.
 * Synchronous: *nonce_out is valid on return. */
int bp_pow_grind(const uint64_t state[12], uint32_t pos, uint32_t bits, uint64_t* nonce_out, void* stream);

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
/* The same for any built-in AIR (bp_stark_prove_synthetic = air_id 0).  air_id 1 .. 7: n_cols = 2431 / 524 / 45 / 309 / 299 / 2414 / 1217,
 * n_const = 0, deg_pow = 1, rate_bits = 1; const_seed is ignored.  The air_id is header word 14 of the proof. */
int bp_stark_prove_air(uint32_t air_id, const bp_stark_cfg* cfg, uint64_t seed, uint64_t const_seed, int device,
                       uint8_t** out, size_t* out_len);
/* The CPU verifier (csrc/verifier.cpp, what VerifierState::verify runs per proof, verifier_state.rs:56-71) on one
 * table proof of bp_stark_prove_air: same transcript prologue.  const_cap: the 2^cap_height x 4 words of the
 * constants commitment when n_const > 0, else NULL.  Host only; BP_ERR_VERIFY + bp_last_error() on rejection. */
int bp_stark_verify_air(uint32_t air_id, const bp_stark_cfg* cfg, const uint64_t* const_cap, const uint8_t* proof,
                        size_t len);
/* AIR 8 (plonk) binds four public inputs to its first row: the hash of the proof's public-input list, which the circuit
 * computes in its hash rows.  A lone table proof (bp_stark_prove_air(8, ...)) has the four-word list
 * bp_stark_public_input_list(seed); bp_stark_public_inputs = its hash, what the verifier is given (pub = NULL: four zeros,
 * what every other AIR has). */
void bp_stark_public_inputs(uint64_t seed, uint64_t out[4]);
void bp_stark_public_input_list(uint64_t seed, uint64_t out[4]);
int bp_stark_verify_air_pub(uint32_t air_id, const bp_stark_cfg* cfg, const uint64_t* const_cap, const uint64_t* pub,
                            const uint8_t* proof, size_t len);
/* bp_stark_prove_synthetic keeps one worker (stream + device arena, up to ~100 GB for a 2^20 x 2432
 * table) parked per device between calls, because re-allocating it costs more than the proof.
 * This frees the parked workers. */
void bp_release_cached_memory(void);
void bp_free_buffer(uint8_t* buf);

/* ------------------------------------------------------------------------------------------
 * L1 -- the reference's public API for this path, byte-for-byte shaped.
 *
 *   reference (Rust)                                              this ABI
 *   ProverStateBuilder::default() / set_<t>_circuit_size / build   bp_config_default, bp_state_build
 *     (prover_state.rs:34-53, 55-75, 80-100; constants.rs:6-18)
 *   generate_txn_proof(&ProverState, TxnProofGenIR, abort)         bp_generate_txn_proof
 *     (proof_gen.rs:39-56)
 *   generate_agg_proof(&ProverState, &lhs, &rhs)                   bp_generate_agg_proof
 *     (proof_gen.rs:61-79; is_agg()/intern()/public_values(): proof_types.rs:55-74)
 *   generate_block_proof(&ProverState, Option<&parent>, &agg)      bp_generate_block_proof
 *     (proof_gen.rs:85-110)
 *   VerifierState::from(&ProverState) / build_verifier / verify    bp_verifier_state_*, bp_verify_block_proof
 *     (verifier_state.rs:34-42, 46-52, 56-71)
 *
 * The zkEVM tables and recursion circuits behind those calls are upstream-only, so
 * the work performed is the synthetic txn proof of SURVEY.md section 8(d): 7 table STARKs (one per
 * AllStark table, positional order arithmetic, byte_packing, cpu, keccak, keccak_sponge, logic,
 * memory -- prover_state.rs:85-93), a 3-deep recursion-shaped chain per table and a root proof;
 * an aggregation / block proof is one recursion-shaped proof bound to its children's digests.
 * Byte formats (the reference fixes none: proof_types.rs:12,25,35,46 only derive serde) are defined
 * in DESIGN.md section 6: little-endian u64 words throughout.
 * ------------------------------------------------------------------------------------------ */
#define BP_NUM_TABLES 7

typedef struct bp_config {
  /* per-table supported log2 trace heights, Range<usize> lo..hi, hi exclusive (constants.rs:6-18) */
  uint32_t table_log_lo[BP_NUM_TABLES], table_log_hi[BP_NUM_TABLES];
  /* StarkConfig::standard_fast_config(): rate_bits 1, cap_height 4, 84 queries, 16 PoW bits,
   * ConstantArityBits(4, 5)  [UPSTREAM-UNVERIFIED values, runtime parameters here] */
  uint32_t stark_rate_bits, stark_cap_height, stark_num_queries, stark_pow_bits, arity_bits, final_poly_bits;
  /* CircuitConfig::standard_recursion_config() shaped proofs: 2^13 rows x 135 wires, rate_bits 3, 28 queries; 84
   * preprocessed constant columns (4 gate constants + 80 sigmas) for the PLONK-shaped circuit, which is the default */
  uint32_t rec_log_n, rec_n_cols, rec_n_const, rec_rate_bits, rec_num_queries, rec_pow_bits;
  uint32_t shrink_depth; /* recursion-shaped proofs per table before the root (3; at least 1: the root circuit walks paths of the recursion shape) */
  uint32_t rec_air_id;   /* what the recursion-shaped proofs are proofs OF: 0 = the synthetic AIR on rec_n_cols x rec_n_const
                          * columns (rounds 1-3; 82 constants); 8 (default) = the PLONK-shaped circuit of AIR 8 (csrc/air.hpp: gates by constants,
                          * public inputs bound in-circuit, copy constraints by the permutation argument), which needs
                          * rec_n_cols = 135 and rec_n_const = 84 -- the proof system of upstream's recursion circuits
                          * (prove_aggregation / prove_block, proof_gen.rs:66-75, 97-103), still not a verifier circuit */
  int32_t device;        /* HIP device index */
  uint32_t n_workers;    /* concurrent provers (one HIP stream + arena each) */
  uint64_t arena_bytes;  /* device arena per worker.  A table height inside its configured range is PROVABLE when
                          * the table's working set -- about 8 * 2^log_n * width * (2 + 2^rate_bits) * 1.3 bytes: values,
                          * coefficients, LDE, digests and FRI scratch -- fits here; otherwise the call returns
                          * BP_ERR_DEVICE ("device arena exhausted").  The default ranges are the reference's
                          * (constants.rs:6-18, up to 2^30 rows), the arena is the deployment's choice: 5-6 GiB
                          * holds the S1 transfer-txn shapes, a 2^20 x 2432 table needs ~100 GiB. */
} bp_config;

typedef struct bp_state bp_state;                   /* ProverState (prover_state.rs:17-20) */
typedef struct bp_verifier_state bp_verifier_state; /* VerifierState (verifier_state.rs:19-23) */

void bp_config_default(bp_config* cfg);
int bp_state_build(const bp_config* cfg, bp_state** out); /* "very expensive call" in the reference */
void bp_state_free(bp_state* s);
uint64_t bp_state_device_bytes(const bp_state* s);
/* What bp_state_build found about the environment the state runs in, as text ("" when there is nothing to say; valid
 * until bp_state_free).  Two things are reported.  (1) RUN-TIME PREREQUISITE: one prover = one HIP stream, and ROCm
 * multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues -- 4 unless the environment variable says
 * otherwise, read once when the HIP runtime starts.  With fewer queues than n_workers the provers share queues and
 * serialise; export GPU_MAX_HW_QUEUES >= n_workers (bench.py: 32) before the process's first HIP call.  (2) The
 * host-wait mode the device was left in (bp_host_wait_mode 2: the library's own poll-and-sleep wait is in use). */
const char* bp_state_warnings(const bp_state* s);
/* the configuration the state was built with */
int bp_state_config(const bp_state* s, bp_config* out);

/* abort_flag: nullable; polled between kernel stages (Option<Arc<AtomicBool>>, proof_gen.rs:42). */
int bp_generate_txn_proof(const bp_state* s, const uint8_t* ir, size_t ir_len, const volatile int32_t* abort_flag,
                          uint8_t** out, size_t* out_len);
/* generate_txn_proof (proof_gen.rs:39-56) for a transaction whose Keccak table attests GIVEN hashing work: the
 * IR must carry the Keccak-AIR flag (bp_ir_set_keccak_air) and keccak_inputs are the states that go into the
 * table's permutations (n_perms x 25 lanes; at most 2^log_n / 24 rounded up; the rest of the table is permutations
 * of the all-zero state).  The decoder side derives them from GenerationInputs with bp_keccak256_permutation_inputs
 * over signed_txn and contract_code (decoding.rs:131-145; block_driver.irs_from_generation_inputs). */
int bp_generate_txn_proof_keccak(const bp_state* s, const uint8_t* ir, size_t ir_len, const uint64_t* keccak_inputs,
                                 size_t n_perms, const volatile uint8_t* abort_flag, uint8_t** out, size_t* out_len);
/* The general form: witness data for any table that has an AIR, instead of a witness drawn from the seed (the decoder
 * side derives it from GenerationInputs: proof_protocol_decoder_amd/block_driver.py).  A table is given data when its
 * has_* field is non-zero (n may be 0: a table of padding only) and its IR flag is set (bp_ir_set_*_air).  Items beyond
 * n are padding: Keccak permutations of the all-zero state, rows without an operation, and for the memory log reads of
 * the last address at later and later times.  Layouts as for the bp_*_trace entry points: keccak_inputs [n][25],
 * logic_ops / arithmetic_ops [n][9], memory_log [n][11] sorted by (address, timestamp), byte_sequences [n][6] (with the address and timestamp of the
 * memory operation each names when the memory table is real too: bp_byte_packing_trace).  When the sponge table and the
 * logic table are both proven by their AIRs the logic table's FIRST rows are not the caller's: rows 5 p + m are the XOR of
 * limbs 8 m .. 8 m + 7 of sponge row p's rate with its block (the lookup keccak_sponge -> logic), for the min(sponge rows,
 * logic rows / 5) sponge rows the table has room for; logic_ops (or seeded operations) follow them, and more operations
 * than fit behind them are BP_ERR_INVALID_INPUT.
 * Given data is CHECKED: the prover does not validate a witness and nothing downstream verifies the table proofs (upstream's
 * root circuit would), so the table proof made from caller-given data is verified on the host before the call goes on;
 * data that does not satisfy the table's AIR (a log that is not a memory, sponge rows that do not chain, ...) returns
 * BP_ERR_VERIFY.  (bp_generate_txn_proof_keccak likewise; every set of permutation inputs satisfies the Keccak-f AIR.) */
typedef struct bp_txn_witness {
  const uint64_t* keccak_inputs;   size_t n_perms;           int has_keccak;
  const uint64_t* logic_ops;       size_t n_logic_ops;       int has_logic;
  const uint64_t* memory_log;      size_t n_memory_ops;      int has_memory;
  const uint64_t* arithmetic_ops;  size_t n_arithmetic_ops;  int has_arithmetic;
  const uint64_t* byte_sequences;  size_t n_byte_sequences;  int has_byte_packing;
  const uint64_t* sponge_rows;     size_t n_sponge_rows;     int has_keccak_sponge;  /* [n][44], bp_keccak256_sponge_rows */
} bp_txn_witness;
int bp_generate_txn_proof_witness(const bp_state* s, const uint8_t* ir, size_t ir_len, const bp_txn_witness* data,
                                  const volatile uint8_t* abort_flag, uint8_t** out, size_t* out_len);
/* What upstream's `prove` yields before the recursion starts (its AllProof; reached from proof_gen.rs:44-52): the seven
 * table proofs of the transaction on their ONE transcript, with the public values and the lookup challenges -- the part of
 * bp_generate_txn_proof that the recursion-shaped proofs then digest.  data: nullable, as for
 * bp_generate_txn_proof_witness.  Bytes (little-endian u64 words): "BPGTABLS", 7, the 13 public values, the four lookup
 * challenges, then per table: air_id, log_n, n_cols, n_words and the table's STARK proof.
 * bp_verify_txn_table_proofs is upstream's verify_proof(all_stark, all_proof, config) on the CPU: every table proof against
 * the shared transcript AND the cross-table lookups between the tables that are proven with their AIRs (csrc/air.hpp,
 * namespace ctl: keccak_sponge -> keccak_f -- what the sponge table hashes is what the Keccak-f table permutes --,
 * byte_packing -> memory -- the word a sequence spells is the memory operation it names): for both challenge sets the looking and the looked running products agree at the first row.  cfg: the STARK parameters only.
 * The same lookup check runs inside every bp_generate_txn_proof* call (upstream's root circuit does it in-circuit): tables
 * that do not form one statement end the call with BP_ERR_VERIFY. */
int bp_generate_txn_table_proofs(const bp_state* s, const uint8_t* ir, size_t ir_len, const bp_txn_witness* data,
                                 const volatile uint8_t* abort_flag, uint8_t** out, size_t* out_len);
int bp_verify_txn_table_proofs(const bp_config* cfg, const uint8_t* table_proofs, size_t len);
/* The same with the STATEMENT fixed by the verifier, as upstream's verify_proof has it (all_stark is the verifier's): ir
 * = the transaction's IR; a table whose header names another AIR, height or width than the IR does (e.g. a Keccak-f
 * table relabelled as synthetic, which would drop its constraints and both of its lookups), or public values other than
 * the IR's, is BP_ERR_VERIFY.  bp_verify_txn_table_proofs alone takes air_id / log_n / n_cols and the public values
 * from the blob, i.e. from the prover: its caller must compare that header with what it expects. */
int bp_verify_txn_table_proofs_for(const bp_config* cfg, const uint8_t* ir, size_t ir_len, const uint8_t* table_proofs,
                                   size_t len);
/* the same call taking the reference's own flag: Arc<AtomicBool> is ONE byte, `flag.as_ptr()` binds here directly */
int bp_generate_txn_proof_u8(const bp_state* s, const uint8_t* ir, size_t ir_len, const volatile uint8_t* abort_flag,
                             uint8_t** out, size_t* out_len);
int bp_generate_agg_proof(const bp_state* s, const uint8_t* lhs, size_t lhs_len, int lhs_is_agg,
                          const uint8_t* rhs, size_t rhs_len, int rhs_is_agg, uint8_t** out, size_t* out_len);
/* parent may be NULL (checkpoint heights, proof_gen.rs:83-84). *b_height = block number of the proof. */
int bp_generate_block_proof(const bp_state* s, const uint8_t* parent, size_t parent_len, const uint8_t* agg,
                            size_t agg_len, uint8_t** out, size_t* out_len, uint64_t* b_height);

int bp_verifier_state_from_prover(const bp_state* s, bp_verifier_state** out);
int bp_verifier_state_build(const bp_config* cfg, bp_verifier_state** out);
/* verifier-only deployment: the three circuit caps (root, agg, block), 4 << stark_cap_height words each */
int bp_verifier_state_from_caps(const bp_config* cfg, const uint64_t* caps, bp_verifier_state** out);
void bp_verifier_state_free(bp_verifier_state* v);
/* CPU only.  BP_ERR_VERIFY with a reason in bp_last_error() when rejected. */
int bp_verify_block_proof(const bp_verifier_state* v, const uint8_t* proof, size_t len);
/* same check for txn / agg containers (not in the reference API; used by tests and the block driver) */
int bp_verify_proof(const bp_verifier_state* v, const uint8_t* proof, size_t len);

/* Serialise a TxnProofGenIR for the synthetic workload; out: BP_IR_WORDS u64. */
#define BP_IR_WORDS 25
#define BP_PV_WORDS 13
int bp_ir_encode(uint64_t block_number, uint64_t txn_number_before, uint64_t gas_used_before,
                 uint64_t gas_used_after, const uint64_t state_root_before[4], uint64_t seed,
                 const uint32_t table_log_n[BP_NUM_TABLES], const uint32_t table_width[BP_NUM_TABLES],
                 uint64_t out_words[BP_IR_WORDS]);
/* A dummy entry (protocol_decoder/src/decoding.rs:484-520, used to pad blocks of 0 or 1 transactions to the two
 * entries an aggregation needs, :304-347): proven like a transaction, but its public values do not
 * advance -- txn_number_after = txn_number_before, gas unchanged, state_root_after = state_root_before. */
int bp_ir_encode_dummy(uint64_t block_number, uint64_t txn_number, uint64_t gas_used, const uint64_t state_root[4],
                       uint64_t seed, const uint32_t table_log_n[BP_NUM_TABLES],
                       const uint32_t table_width[BP_NUM_TABLES], uint64_t ir_out[BP_IR_WORDS]);
/* Marks an encoded IR (flag 0x100 of the version word) so that its Keccak table -- table index 3 in the positional
 * order of prover_state.rs:85-93 -- is proven with the Keccak-f[1600] AIR (air_id 1: 2431 columns, the witness is
 * ceil(2^log_n / 24) permutations drawn from the seed) instead of the synthetic AIR.  The table's width must be 2431. */
int bp_ir_set_keccak_air(uint64_t ir[BP_IR_WORDS], int on);
/* The same for the logic table (flag 0x200; table index 5): proven with the logic AIR (air_id 2: 524 columns, one
 * operation per row drawn from the seed).  The table's width must be 524. */
int bp_ir_set_logic_air(uint64_t ir[BP_IR_WORDS], int on);
/* ... and for the memory table (flag 0x400; table index 6): the memory AIR (air_id 3: 45 columns, a sorted log drawn
 * from the seed).  The table's width must be 45. */
int bp_ir_set_memory_air(uint64_t ir[BP_IR_WORDS], int on);
/* ... and for the arithmetic table (flag 0x800; table index 0): the arithmetic AIR (air_id 4: 309 columns). */
int bp_ir_set_arithmetic_air(uint64_t ir[BP_IR_WORDS], int on);
/* ... or, instead (one AIR per table), with the multiplication AIR (flag 0x4000; air_id 7: 1217 columns, x * y = z + 2^256 w
 * per row; arithmetic_ops then are [n][9] = is_mul, the words of x, the words of y).  The table's width must be 1217. */
int bp_ir_set_arithmetic_mul_air(uint64_t ir[BP_IR_WORDS], int on);
/* ... and for the byte-packing table (flag 0x1000; table index 1): the byte-packing AIR (air_id 5: 299 columns). */
int bp_ir_set_byte_packing_air(uint64_t ir[BP_IR_WORDS], int on);
/* ... and for the Keccak sponge table (flag 0x2000; table index 4): the Keccak sponge AIR (air_id 6: 2414 columns). */
int bp_ir_set_keccak_sponge_air(uint64_t ir[BP_IR_WORDS], int on);
/* public values of a proof container: txn_before, txn_after, gas_before, gas_after, root_before[4],
 * root_after[4], block_number */
int bp_proof_public_values(const uint8_t* proof, size_t len, uint64_t pv_out[BP_PV_WORDS], int* kind_out);

/* ------------------------------------------------------------------------------------------
 * Next row: Erigon compact block-witness decoder + state-trie root
 * (protocol_decoder/src/compact/compact_prestate_processing.rs:1240-1281 process_compact_prestate,
 * compact_to_partial_trie.rs:37-165).  Host-only (it is CPU parsing in the reference too).
 * Pinned by the reference's own golden vectors (complex_test_payloads.rs:14-30 and
 * compact_prestate_processing.rs:1439,1483-1492).  Any out-pointer may be NULL.
 * n_accounts_missing_storage counts account leaves whose non-empty storage root has no extracted
 * storage trie (the invariant the reference's test harness asserts, complex_test_payloads.rs:73-90).
 * ------------------------------------------------------------------------------------------ */
int bp_compact_decode(const uint8_t* witness, size_t len, uint8_t* header_version, uint8_t state_root[32],
                      uint32_t* n_accounts, uint32_t* n_storage_tries, uint32_t* n_code,
                      uint32_t* n_accounts_missing_storage);
/* The reference's FULL output of the same call (ProcessedCompactOutput{header, witness_out{tries{state,
 * storage}, code}}, compact_prestate_processing.rs:1243-1281), release with bp_free_buffer:
 *   "BPGCWIT1" | version:u8 | state: len:u32 trie | n_storage:u32 (hashed_addr[32] len:u32 trie)* |
 *   n_code:u32 (code_hash[32] len:u32 bytes)*
 * Storage tries are keyed by HASHED ACCOUNT ADDRESS (the re-keying of compact_to_partial_trie.rs:167-190), lists
 * in ascending key order, integers little-endian.  A trie is its nodes in preorder, one tag byte each:
 *   0x00 empty | 0x01 hash[32] | 0x02 mask:u16 vlen:u32 value child* (present children, nibble order) |
 *   0x03 nkey:u8 nibble[nkey] child | 0x04 nkey:u8 nibble[nkey] vlen:u32 value
 * (leaf values are what the reference inserts: rlp(AccountRlp) for accounts, rlp(value) for storage slots). */
int bp_compact_decode_full(const uint8_t* witness, size_t len, uint8_t** out, size_t* out_len);
/* one instruction per text line; release with bp_free_buffer */
int bp_compact_instructions(const uint8_t* witness, size_t len, uint8_t** text_out, size_t* text_len);
void bp_keccak256(const uint8_t* data, size_t len, uint8_t out[32]);
/* Keccak-256 of `data` and the 25-lane state (lane index x + 5y) that goes into each of its permutations
 * (len / 136 + 1 of them): the rows the Keccak table of a transaction that hashes `data` has to contain.
 * states_out (room for max_perms x 25 words) and digest_out may be NULL; *n_perms_out is always set.  Host only. */
int bp_keccak256_permutation_inputs(const uint8_t* data, size_t len, uint8_t digest_out[32], uint64_t* states_out,
                                    size_t max_perms, size_t* n_perms_out);
/* The same hash as rows of the Keccak sponge table (bp_keccak_sponge_trace): 44 words per 136-byte block.  rows_out
 * may be NULL to count. */
int bp_keccak256_sponge_rows(const uint8_t* data, size_t len, uint8_t digest_out[32], uint64_t* rows_out, size_t max_rows,
                             size_t* n_rows_out);

/* ------------------------------------------------------------------------------------------
 * Next row: the txn IR producer, BlockTrace::into_txn_proof_gen_ir
 * (protocol_decoder/src/processed_block_trace.rs:38-50, 210-332; decoding.rs:81-177, 179-292, 304-347, 356-428):
 * compact pre-image + per-txn account traces -> one GenerationInputs per transaction (minimal partial tries,
 * deltas replayed over the block's trie state, roots after), padded to >= 2 entries with dummies, withdrawals on
 * the last dummy.  Host-only, sequential by nature.  Unpinned by the reference (it has no test for this path);
 * pinned here by invariants (tests/test_decoding.py).  All integers little-endian, U256 as 32 bytes big-endian.
 *
 * in  = "BPGTRAC1" | witness: len:u32 bytes | n_txn:u32 txn* | checkpoint_state_trie_root[32] |
 *       block_metadata: len:u32 bytes | block_hashes: len:u32 bytes (both opaque upstream types, copied through) |
 *       n_withdrawals:u32 (address[20] amount[32])* | n_code:u32 (code_hash[32] len:u32 bytes)*  (CodeHashResolveFunc as a table)
 * txn = n_traces:u32 trace* | byte_code: len:u32 bytes | new_txn_trie_node_byte | new_receipt_trie_node_byte | gas_used:u64
 * trace = address[20] flags:u8 [balance[32]] [nonce[32]] [n:u32 slot[32]*] [n:u32 (slot[32] value[32])*] [code_hash[32] | len:u32 code]
 *       flags: 1 balance, 2 nonce, 4 storage_read, 8 storage_written, 16 code_usage read, 32 code_usage write, 64 self_destructed
 * out = "BPGGENI1" | n:u32 ir* | final_state_root[32]
 * ir  = txn_number_before[32] gas_used_before[32] gas_used_after[32] | has_signed_txn:u8 signed_txn: len:u32 bytes |
 *       n:u32 (address[20] amount[32])* withdrawals | tries: state, transactions, receipts (each len:u32 trie),
 *       n:u32 (hashed_addr[32] len:u32 trie)* storage | trie_roots_after: state[32] transactions[32] receipts[32] |
 *       checkpoint_state_trie_root[32] | n:u32 (code_hash[32] len:u32 bytes)* contract_code | block_metadata | block_hashes
 * Release with bp_free_buffer.
 * ------------------------------------------------------------------------------------------ */
int bp_decode_block_trace(const uint8_t* trace, size_t len, uint8_t** out, size_t* out_len);

/* ------------------------------------------------------------------------------------------
 * GenerationInputs as the prover's input (csrc/gi.cpp).  The reference's generate_txn_proof consumes
 * TxnProofGenIR = GenerationInputs (plonky_block_proof_gen/src/proof_gen.rs:39-43, protocol_decoder/src/types.rs:48,
 * fields as populated at decoding.rs:131-145): ONE entry of what BlockTrace::into_txn_proof_gen_ir emits.  Here: one
 * entry of a "BPGGENI1" buffer (bp_decode_block_trace; a host holding a single GenerationInputs wraps it as a
 * one-entry buffer).  The library derives the 25-word IR -- txn number, gas and the state-root public
 * value threaded entry to entry (bp_gi_chain: what the entries before left; decoding.rs:106-154 threads the same
 * three), the witness seed = Keccak-256(signed_txn | trie roots after | withdrawals) -- and, for the tables proven
 * with their AIRs, the witness of the entry's OWN hashing work: Keccak-256 of signed_txn and of every contract_code
 * entry (and of the hash-referenced nodes of its partial tries) as Keccak-f permutations, sponge rows, and the
 * memory log / byte-packing sequences of the same bytes (AIRS.md section 3).  Table heights grow to hold that work.
 * ------------------------------------------------------------------------------------------ */
#define BP_GI_KECCAK_AIR 1u         /* table 3 proven with the Keccak-f AIR over the entry's hashing work */
#define BP_GI_KECCAK_TRIE_NODES 2u  /* ... which includes the hashing of the entry's partial tries */
#define BP_GI_MEMORY_AIR 4u         /* table 6: the memory log of the hashed bytes */
#define BP_GI_BYTE_PACKING_AIR 8u   /* table 1: the byte-packing sequences of the hashed bytes */
#define BP_GI_KECCAK_SPONGE_AIR 16u /* table 4: the sponge rows absorbing the same strings */
#define BP_GI_LOGIC_AIR 32u         /* table 5: the logic AIR, holding the sponge rows' XORs first (keccak_sponge -> logic; needs the sponge flag) */
typedef struct bp_gi_options {
  uint64_t block_number;
  uint32_t table_log_n[BP_NUM_TABLES];  /* base heights (the tables that hold given work grow beyond them as needed) */
  uint32_t table_width[BP_NUM_TABLES];  /* widths of the tables that stay synthetic */
  uint32_t flags;                       /* BP_GI_* */
} bp_gi_options;
typedef struct bp_gi_chain {            /* where an entry starts: what the entries before it left */
  uint64_t txn_number, gas_used, state_root[4];
} bp_gi_chain;
int bp_gi_count(const uint8_t* geni, size_t len, uint32_t* n_entries);
/* chain before entry 0: counters 0, state root = the first entry's state-trie root folded into four field elements */
int bp_gi_chain_start(const uint8_t* geni, size_t len, bp_gi_chain* chain);
/* the IR of entry `entry` given the chain before it (heights as the prover will use them); *chain moves past the entry */
int bp_gi_entry_ir(const uint8_t* geni, size_t len, uint32_t entry, const bp_gi_options* opt, bp_gi_chain* chain,
                   uint64_t ir_out[BP_IR_WORDS]);
/* generate_txn_proof(&ProverState, GenerationInputs, Option<Arc<AtomicBool>>): entry `entry`, the chain before it in
 * *chain (moved past the entry on success).  abort_flag as in bp_generate_txn_proof_u8. */
int bp_generate_txn_proof_gi(const bp_state* s, const uint8_t* geni, size_t len, uint32_t entry, const bp_gi_options* opt,
                             bp_gi_chain* chain, const volatile uint8_t* abort_flag, uint8_t** out, size_t* out_len);

/* ------------------------------------------------------------------------------------------
 * The shard scheduler (csrc/gi.cpp): all txn proofs of a CONTIGUOUS slice (aggregation needs contiguous ranges,
 * proof_types.rs:23-24) and its aggregation tree on a pool of threads; every aggregation starts the moment both of its
 * children exist and goes AHEAD of the transactions still waiting for a thread.  The reference leaves this to its
 * scheduler (docs/usage_seq_diagrams.md:8-20); bench.py's headline rate is measured through bp_prove_shard.
 * bp_aggregation_plan: entry k = node n + k = (left, right) node ids, leaves are 0..n-1, the last entry is the root;
 *   shape 0 = balanced (adjacent pairs level by level, an odd tail carried up), 1 = pairs then a left-to-right chain.
 *   pairs: room for 2 (n - 1) ids or NULL; returns n - 1, or UINT32_MAX for n = 0 / an unknown shape.
 * bp_run_shard: the scheduler over caller-supplied work (how the CPU tests drive it, and how a host with another
 *   prover behind the same tree would): leaf(ctx, i) makes leaf i, agg(ctx, ...) merges two children; buffers are
 *   malloc()ed by the callee and owned by the scheduler.  root_out: the slice's one proof (the leaf itself when
 *   n = 1).  leaf_out / leaf_len: NULL, or room for n pointers / lengths -- the leaves are then handed to the caller
 *   too (each released with bp_free_buffer).  The first failing node ends the call with its status and message.
 * n_threads = 0: the state's n_workers (bp_run_shard: 1). */
typedef struct bp_shard_options {
  uint32_t n_threads;
  uint32_t tree_shape;
} bp_shard_options;
typedef int (*bp_shard_leaf_fn)(void* ctx, uint32_t index, uint8_t** out, size_t* out_len);
typedef int (*bp_shard_agg_fn)(void* ctx, const uint8_t* lhs, size_t lhs_len, int lhs_is_agg, const uint8_t* rhs, size_t rhs_len,
                               int rhs_is_agg, uint8_t** out, size_t* out_len);
uint32_t bp_aggregation_plan(uint32_t n, uint32_t shape, uint32_t* pairs);
int bp_run_shard(uint32_t n, const bp_shard_options* opt, bp_shard_leaf_fn leaf, bp_shard_agg_fn agg, void* ctx,
                 const volatile uint8_t* abort_flag, uint8_t** root_out, size_t* root_len, uint8_t** leaf_out, size_t* leaf_len);
/* irs: n IRs, ir_stride bytes apart (>= BP_IR_WORDS * 8) */
int bp_prove_shard(const bp_state* s, const uint8_t* irs, size_t ir_stride, uint32_t n, const bp_shard_options* opt,
                   const volatile uint8_t* abort_flag, uint8_t** root_out, size_t* root_len, uint8_t** txn_out, size_t* txn_len);
/* n proofs made elsewhere (txn or aggregation proofs, contiguous in this order: the sub-block proofs gathered from the
 * other ranks) folded into one along the same plan */
int bp_aggregate_proofs(const bp_state* s, const uint8_t* const* proofs, const size_t* lens, uint32_t n, const bp_shard_options* opt,
                        uint8_t** out, size_t* out_len);
/* entries [first, first + n) of a decoded block */
int bp_prove_shard_gi(const bp_state* s, const uint8_t* geni, size_t len, uint32_t first, uint32_t n, const bp_gi_options* gi,
                      const bp_shard_options* opt, const volatile uint8_t* abort_flag, uint8_t** root_out, size_t* root_len,
                      uint8_t** txn_out, size_t* txn_len);

/* state root after one synthetic txn (host-side helper for building a chain of IRs) */
int bp_state_root_after(const uint64_t root_before[4], uint64_t seed, uint64_t txn_number, uint64_t out[4]);

/* Optional HIP-event timing of kernel families on their own streams (bench.py roofline leg).
 * family 0: LDE coset NTT (LDS-resident DIT), 1: inverse NTT (DIF); total_alg_bytes uses the
 * algorithmic byte counts of SURVEY.md section 8(d).  family 2: Merkle leaf hashing (integer-ALU
 * bound): the third output counts Poseidon PERMUTATIONS instead of bytes.  The other HBM-class kernels of
 * section 8(d): 3 FRI fold (16 M (1 + 1/16)), 4 openings (8 n C), 5 FRI alpha-combination (8 n C), 6 auxiliary
 * running products (8 n per column read or written), 7 + air_id the quotient kernel K5 of that AIR (every LDE
 * element read once, two quotient columns written). */
void bp_profile_enable(int on);
void bp_profile_reset(void);
int bp_profile_read(int family, uint64_t* launches, double* total_ms, double* total_alg_bytes);

#ifdef __cplusplus
}
#endif
#endif /* BPG_H */
