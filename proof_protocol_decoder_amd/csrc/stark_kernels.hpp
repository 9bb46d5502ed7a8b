// stark_kernels.hpp -- argument blocks and launchers of the per-table STARK kernels
// (stark_kernels.hip) plus the NTT / Merkle entry points used by the host prover.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "air.hpp"
#include "gl.hpp"

namespace bpg {

struct Ctl {
  uint64_t v[4];  // beta0, gamma0, beta1, gamma1 (grand-product challenge sets)
  // the table's public inputs (AIR 8: the four words of the hash of the proof's public-input list, bound to the first
  // row in-circuit); zero for the tables that have none
  uint64_t pub[4] = {0, 0, 0, 0};
};

// Proofs of ONE shape proved in lock-step (the seven per-table recursion chains of a transaction,
// proofgen.cpp): every kernel below takes the argument blocks of up to MAX_BATCH proofs and runs proof
// blockIdx.z of them; a lone proof is a batch of one.  The blocks travel as kernel arguments (<= 4 KiB).
constexpr uint32_t MAX_BATCH = 8;
template <class A>
struct BatchOf {
  A a[MAX_BATCH];
};
template <class A>
inline BatchOf<A> batch_of(const A* a, uint32_t n) {
  BatchOf<A> b{};
  for (uint32_t i = 0; i < n && i < MAX_BATCH; i++) b.a[i] = a[i];
  return b;
}

struct QuotArgs {
  const uint64_t *trace_lde, *aux_lde, *const_lde;
  uint64_t trace_stride, aux_stride, const_stride;
  uint64_t *partial, *qvals;  // partial: [wg rows][2][rows], only when the units are spread over grid.y
  const uint64_t* tw_n;       // w_n^e, e < n/2
  const uint64_t* apow;       // [2][n_constraints] alpha_j^e, then 48 per-coset words; filled by launch_quotient
  uint32_t air_id, log_n, rate_bits, n_cols, n_const, n_aux, deg_pow;
  // the constraint list (air.hpp): AIR constraints, then two per aux column; units = AIR units, then CTL units
  uint32_t n_air_constraints, n_constraints, n_air_units, n_ctl_units, aux_per_unit, units_per_wg;
  // AIR 8: its Poseidon gate (unit 10) is a pass of its own (quotient_plonk_hash_kernel: another register budget), whose
  // sums are one more row of `partial`; n_air_units then counts the ten chunk units only
  uint32_t side_rows;
  uint64_t alpha0, alpha1, g, g_inv, n_inv;
  Ctl ctl;
};
struct QuotCoset {  // per coset: 7*w_M^t, g_t^n - 1 and its inverse (the same for every proof of a shape)
  uint64_t g_t[16], zh_t[16], zh_inv_t[16];
};
struct SynthTraceArgs {
  uint64_t* trace;
  const uint64_t* consts;
  uint64_t seed;
};
struct AuxArgs {
  const uint64_t* trace;
  uint64_t* aux;
  Ctl ctl;
  const uint64_t* consts = nullptr;  // AIR 8: the preprocessed constant columns on the trace domain ([K][n]: the sigmas)
};
struct PowerVecArgs {
  uint64_t* out;  // n_points vectors of 2n words each: point y at out + y * 2n
  gl::Ext z[3];
};
struct AlphaPowArgs {
  uint64_t* out;
  gl::Ext alpha;
};
struct CombineReduceArgs {
  const uint64_t* partial;
  uint64_t* g;
};
struct ChunkArgs {
  const uint64_t* e;          // [2][2^r][n] per-coset inverse NTT output (bit-reversed)
  const uint64_t* inv_scale;  // [2^r][n]: g_t^(-bitrev(pos))
  uint64_t* out;              // [2*2^r] chunk coefficient columns
  uint64_t out_stride;
  uint32_t log_n, rate_bits;
  uint64_t wr_inv_pow[16];    // w_{2^r}^(-k)
  uint64_t chunk_scale[16];   // (7^n)^(-n1) / 2^r
};
struct CombineArgs {
  const uint64_t* coeffs;
  uint64_t stride;
  uint32_t log_n, n_cols, cols_per_chunk, chunk_base;
  int32_t exp_base[3];         // alpha exponent of this oracle's first column per batch; <0: not in batch
  const uint64_t* alpha_pows;  // ext pairs alpha^j
  uint64_t* partial;           // [chunks][6][n]
};
struct CombineMulti {  // the oracles of one proof; a[o].chunk_base = first global chunk of oracle o (ascending)
  CombineArgs a[4];
  uint32_t n_oracles;
};
struct OpenMulti {  // up to five opening sets in one launch: segment s covers blocks [first_col[s], first_col[s + 1])
  const uint64_t* coeffs[5];
  const uint64_t* pw[5];
  uint64_t* out[5];
  uint32_t first_col[6], n_points[5];
  uint64_t stride;
  uint32_t log_n, n_segs;
};
struct FriInitArgs {
  const uint64_t* glde;  // [6][rows]
  uint64_t* out;         // ext AoS [rows]
  const uint64_t* tw_n;
  uint32_t log_n, rate_bits;
  uint64_t g_t[16];
  gl::Ext y[3], z[3], alpha_shift[3];
};
struct FriLayerArgs {
  const uint64_t* values;  // ext AoS, coset-major [2^r][n_l]
  uint64_t* out;           // folded layer, ext AoS [2^r][n_l/arity]
  uint64_t* digests;       // leaf digests in leaf order
  const uint64_t* tw_nl_inv;
  uint32_t log_nl, rate_bits, arity_bits;
  uint64_t g_t_inv[16];
  uint64_t wa_inv_pow[16];
  uint64_t arity_inv;
  gl::Ext beta;
};
struct PowArgs {
  uint64_t state[12];
  uint64_t base;
  uint32_t pos, bits;
};
struct QueryOracle {
  const uint64_t *lde, *digests;
  uint64_t stride;
  uint32_t n_cols, out_offset;
};
// Query openings of a batch: query q of the launch belongs to proof q / n_queries (MAX_BATCH_QUERIES indices in all)
constexpr uint32_t MAX_BATCH_QUERIES = 256;
struct QueryProof {
  uint64_t* out;
  QueryOracle oracle[4];
  uint64_t* first_leaf = nullptr;  // (nullable) 4 words: the digest of the leaf query 0 opens in oracle `leaf_oracle` (the trace)
};
struct QueryArgs {
  uint64_t x_index[MAX_BATCH_QUERIES];
  uint64_t query_words;
  uint32_t log_n, rate_bits, cap_height, n_queries;  // n_queries: per proof
  uint32_t leaf_oracle = 0;
  QueryProof proof[MAX_BATCH];
};
struct QueryLayer {
  const uint64_t *values, *digests;
  uint32_t log_nl, out_offset;
};
constexpr uint32_t MAX_FRI_LAYERS = 8;  // check_cfg (prover.cpp)
struct QueryLayerProof {
  uint64_t* out;
  QueryLayer layer[MAX_FRI_LAYERS];
};
struct QueryLayerArgs {
  uint64_t x_index[MAX_BATCH_QUERIES];
  uint64_t query_words;
  uint32_t rate_bits, cap_height, arity_bits, n_queries;
  QueryLayerProof proof[MAX_BATCH];
};

// stark_kernels.hip.  Launchers that take `const X* a, uint32_t batch` run `batch` proofs of one shape in one launch.
int launch_synth_constants(uint64_t* d_out, uint32_t log_n, uint32_t n_const, uint64_t seed, hipStream_t st);
int launch_synth_trace(const SynthTraceArgs* a, uint32_t batch, uint32_t log_n, uint32_t n_cols, uint32_t n_const,
                       uint32_t deg_pow, hipStream_t st);
inline int launch_synth_trace(uint64_t* d_trace, const uint64_t* d_consts, uint32_t log_n, uint32_t n_cols,
                              uint32_t n_const, uint32_t deg_pow, uint64_t seed, hipStream_t st) {
  const SynthTraceArgs a{d_trace, d_consts, seed};
  return launch_synth_trace(&a, 1, log_n, n_cols, n_const, deg_pow, st);
}
// AIR 1 witness: n rows x 2430 columns; d_inputs [ceil(n / 24)][25] lanes or null (then drawn from seed)
int launch_keccak_trace(uint64_t* d_trace, const uint64_t* d_inputs, uint32_t log_n, uint64_t seed, hipStream_t st);
int launch_logic_trace(uint64_t* d_trace, const uint64_t* d_inputs, uint32_t log_n, uint64_t seed, hipStream_t st);
int launch_memory_trace(uint64_t* d_trace, const uint64_t* d_inputs, uint32_t log_n, uint64_t seed, hipStream_t st);
int launch_arithmetic_trace(uint64_t* d_trace, const uint64_t* d_inputs, uint32_t log_n, uint64_t seed, hipStream_t st);
int launch_byte_packing_trace(uint64_t* d_trace, const uint64_t* d_inputs, uint32_t log_n, uint64_t seed, hipStream_t st);
// row_limit: rows of a SEEDED table from this one on are padding rows (a table given by the caller is taken as it is)
int launch_keccak_sponge_trace(uint64_t* d_trace, const uint64_t* d_inputs, uint32_t log_n, uint64_t seed, hipStream_t st,
                               uint32_t row_limit = ~0u);
// [n_perms][25] input lanes for a seeded Keccak-f table whose sponge table (trace on the device) is real
int launch_keccak_inputs_from_sponge(const uint64_t* d_sponge_trace, uint32_t sponge_log_n, uint64_t* d_inputs, uint32_t n_perms,
                                     uint64_t seed, hipStream_t st);
// the memory log ([n_mem][11], for launch_memory_trace) that goes with a byte-packing trace: two operations per packing row
// the logic table's operations when the sponge table is proven by its AIR too (keccak_sponge -> logic): five per covered sponge row,
// then the caller's (d_given, may be null) or seeded ones
int launch_logic_inputs_from_sponge(const uint64_t* d_sponge_trace, uint32_t sponge_log_n, uint32_t covered, const uint64_t* d_given,
                                    uint32_t n_given, uint64_t* d_inputs, uint32_t n_logic, uint64_t seed, hipStream_t st);
int launch_memory_inputs_from_byte_packing(const uint64_t* d_pack_trace, uint32_t pack_log_n, uint64_t* d_log, uint32_t n_mem,
                                           hipStream_t st);
// The filter column of a LOOKED table's trace (air::ctl; keccak::COL_G, memory::COL_G), written after its witness and
// before its commitment: Keccak-f table: permutation p is exposed when flag_a[p] + flag_b[p] != 0 (the two flag columns of
// the sponge table's trace, n_flags rows of it); memory table: flag_a = the byte-packing table's trace of n_flags rows
// (its address and timestamp columns name the operations it looks up), flag_b unused.  flag_a null: nothing is exposed.
int launch_lookup_filter(uint32_t air_id, uint64_t* d_trace, uint32_t log_n, const uint64_t* flag_a, const uint64_t* flag_b,
                         uint32_t n_flags, hipStream_t st);
int launch_arithmetic_mul_trace(uint64_t* d_trace, const uint64_t* d_inputs, uint32_t log_n, uint64_t seed, hipStream_t st);
int launch_aux(const AuxArgs* a, uint32_t batch, uint32_t air_id, uint32_t n_cols, uint32_t log_n, hipStream_t st);
// AIR 8 (plonk): the constants (selectors, gate constants, sigmas of the fixed circuit) and the witness
// lay: the public-input list the circuit hashes in its hash rows and the Merkle paths it walks (air::plonk::Layout)
int launch_plonk_constants(uint64_t* d_out, uint32_t log_n, uint64_t seed, const air::plonk::Layout& lay, hipStream_t st);
struct PlonkTraceArgs {
  uint64_t* trace;
  const uint64_t* consts;
  uint64_t seed, pub[4];        // pub = hash_no_pad(the public-input list) = the first output words of the last hash row
  // device-visible [HASH_ROWS_MAX + n_merkle_rows][air::plonk::H_WIRES]: the witness of that hash in the first
  // n_hash_rows rows (poseidon_hash_rows), of the Merkle paths from row HASH_ROWS_MAX on (poseidon_merkle_rows), of the
  // paths' leaf sponges right after them (poseidon_hash_rows of each opened row); n_merkle_rows counts both
  const uint64_t* hash_rows;
  uint32_t n_hash_rows;
  uint32_t n_merkle_rows = 0, arith_row0 = (air::plonk::MERKLE_ROW0 + 3) & ~3u;  // air::plonk::arith_row0(the circuit's layout); default: no paths
};
int launch_plonk_trace(const PlonkTraceArgs* a, uint32_t batch, uint32_t log_n, hipStream_t st);
// every proof of the batch has the shape and the unit spreading of q[0]
int launch_quotient(const QuotArgs* q, uint32_t batch, const QuotCoset& coset, hipStream_t st);
inline int launch_quotient(const QuotArgs& q, const QuotCoset& coset, hipStream_t st) { return launch_quotient(&q, 1, coset, st); }
int launch_quotient_chunks(const ChunkArgs* c, uint32_t batch, hipStream_t st);
int launch_power_vectors(const PowerVecArgs* a, uint32_t batch, uint32_t log_n, uint32_t n_points, hipStream_t st);
// d_out: n_points (<= 3) vectors of 2n words each: point y at d_out + y * 2n
inline int launch_power_vectors(uint64_t* d_out, uint32_t log_n, gl::Ext z0, gl::Ext z1, uint32_t n_points,
                                hipStream_t st, gl::Ext z2 = gl::Ext{1, 0}) {
  const PowerVecArgs a{d_out, {z0, z1, z2}};
  return launch_power_vectors(&a, 1, log_n, n_points, st);
}
int launch_alpha_pows(const AlphaPowArgs* a, uint32_t batch, uint32_t count, hipStream_t st);
int launch_openings(const uint64_t* d_coeffs, uint64_t stride, uint32_t log_n, uint32_t n_cols,
                    const uint64_t* d_pw, uint32_t n_points, uint64_t* d_out, hipStream_t st);
int launch_openings_multi(const OpenMulti* m, uint32_t batch, hipStream_t st);
int launch_combine_partial_multi(const CombineMulti* m, uint32_t batch, uint32_t total_chunks, hipStream_t st);
int launch_combine_all(const CombineMulti* m, uint32_t batch, uint64_t* const* d_g, hipStream_t st);  // one pass, straight into g[6][n]
int launch_combine_reduce(const CombineReduceArgs* a, uint32_t batch, uint32_t n_chunks, uint32_t log_n, hipStream_t st);
int launch_fri_init(const FriInitArgs* a, uint32_t batch, hipStream_t st);
int launch_fri_layer_leaves(const FriLayerArgs* a, uint32_t batch, hipStream_t st);
int launch_fri_fold(const FriLayerArgs* a, uint32_t batch, hipStream_t st);
inline int launch_fri_fold(const FriLayerArgs& a, hipStream_t st) { return launch_fri_fold(&a, 1, st); }
// d_result: one word per proof of the batch (all ones = no witness yet)
int launch_pow(const PowArgs* a, uint32_t batch, uint32_t n_candidates, unsigned long long* d_result, hipStream_t st);
inline int launch_pow(const PowArgs& a, uint32_t n_candidates, unsigned long long* d_result, hipStream_t st) {
  return launch_pow(&a, 1, n_candidates, d_result, st);
}
int launch_query_initial(const QueryArgs& a, uint32_t batch, uint32_t n_oracles, hipStream_t st);
int launch_query_layers(const QueryLayerArgs& a, uint32_t batch, uint32_t n_layers, hipStream_t st);

// ntt.hip
int init_ntt_kernels();
int get_table(int kind, uint32_t log_n, uint32_t rate_bits, const uint64_t** out);  // 0 fwd, 1 inv, 2 coset, 3 inv coset
int intt_nat2br(const uint64_t* in, uint64_t in_stride, uint64_t* out, uint64_t out_stride, uint32_t log_n,
                uint32_t n_cols, bool inverse, hipStream_t st);
int ntt_br2nat(const uint64_t* in, uint64_t in_stride, uint64_t* out, uint64_t out_stride, uint64_t coset_stride,
               uint32_t log_n, uint32_t n_cols, uint32_t n_cosets, const uint64_t* scale, bool inverse,
               hipStream_t st);
// hash_kernels.hip
// the current device's image of the grouped-Poseidon operand tables and the number of groups (2 or 3) it is laid out
// for, or nullptr (knob off)
const uint32_t* group_tables(int* n_groups);
// batch > 1: `batch` trees of one shape in every launch; tree b's matrix / digest buffer / cap mirror sits
// b * lde_bstride / b * dig_bstride / b * (4 << cap_height) words behind the first one's
int merkle_upper_levels(uint64_t* d_digests, uint32_t log_leaves, uint32_t cap_height, hipStream_t st,
                        uint64_t* mirror, bool* mirrored, uint32_t batch = 1, uint64_t dig_bstride = 0);
int merkle_commit_cols(const uint64_t* d_lde, uint64_t lde_stride, uint32_t n_cols, uint32_t log_n, uint32_t rate_bits,
                       uint32_t cap_height, uint64_t* d_digests, hipStream_t st, uint64_t* mirror, bool* mirrored,
                       uint32_t batch = 1, uint64_t lde_bstride = 0, uint64_t dig_bstride = 0);

}  // namespace bpg
