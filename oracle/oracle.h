/* oracle/oracle.h -- C API of the CPU restatement.  TEST INFRASTRUCTURE ONLY (see gl.h header).
 * "parity unpinned" by the reference: plonky_block_proof_gen has no tests; the
 * arithmetic follows the published plonky2 algorithms reached from proof_gen.rs:44-52.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include "gl.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- K1/K2: field + NTT (plonky2_field fft conventions, natural order in and out) ---------- */
void orc_dft_naive(const gl_t* in, gl_t* out, unsigned log_n, int inverse);
void orc_ntt(gl_t* a, unsigned log_n);                 /* values[i] = sum_j a[j] w^(ij) */
void orc_intt(gl_t* a, unsigned log_n);                /* inverse, includes 1/n */
void orc_coset_ntt(gl_t* a, unsigned log_n, gl_t shift);   /* evaluate on shift*<w> */
void orc_coset_intt(gl_t* a, unsigned log_n, gl_t shift);
/* column-major batch: col c at cols + c*stride; uses OpenMP over columns */
void orc_ntt_batch(gl_t* cols, unsigned log_n, size_t n_cols, size_t stride, int inverse);
/* values (n) -> coeffs (n) -> LDE evaluations on 7*<w_{n*2^r}> (natural order), column-major.
 * PolynomialBatch::from_values equivalent.  coeffs_out may be NULL. */
void orc_lde_batch(const gl_t* values, gl_t* coeffs_out, gl_t* lde_out, unsigned log_n,
                   unsigned rate_bits, size_t n_cols, int from_coeffs);

/* ---------- K3/K4: Poseidon + Merkle ---------- */
void orc_poseidon(gl_t state[12]);
void orc_poseidon_batch(gl_t* states, size_t n);
void orc_hash_no_pad(const gl_t* in, size_t len, gl_t out[4]);
void orc_hash_or_noop(const gl_t* in, size_t len, gl_t out[4]);
void orc_two_to_one(const gl_t l[4], const gl_t r[4], gl_t out[4]);
/* Merkle tree over n_leaves = 2^log_leaves leaves.  Leaf k is the row bitrev(k) of the column-major
 * matrix (n_cols columns, stride elements apart) when bitrev_rows != 0, else row k.
 * digests_out: all levels, level 0 (leaves) first, down to the cap level: (2*n_leaves - 2^cap_h)*4
 * words; the last 2^cap_h*4 words are the cap. */
size_t orc_merkle_digest_words(unsigned log_leaves, unsigned cap_h);
void orc_merkle_commit(const gl_t* cols, size_t stride, size_t n_cols, unsigned log_leaves,
                       unsigned cap_h, int bitrev_rows, gl_t* digests_out);
/* row-major leaves (leaf k = leaf_len consecutive words) -- used for FRI layers */
void orc_merkle_commit_rows(const gl_t* leaves, size_t leaf_len, unsigned log_leaves, unsigned cap_h,
                            gl_t* digests_out);
void orc_merkle_path(const gl_t* digests, unsigned log_leaves, unsigned cap_h, size_t leaf,
                     gl_t* path_out /* (log_leaves-cap_h)*4 */);
int orc_merkle_verify(const gl_t* leaf_data, size_t leaf_len, size_t leaf, const gl_t* path,
                      unsigned log_leaves, unsigned cap_h, const gl_t* cap);

/* ---------- K7: challenger (plonky2::iop::challenger duplex sponge) ---------- */
typedef struct {
  gl_t state[12];
  gl_t in[8]; unsigned n_in;
  gl_t out[8]; unsigned n_out;
} orc_challenger;
void orc_ch_init(orc_challenger* c);
void orc_ch_observe(orc_challenger* c, gl_t e);
void orc_ch_observe_many(orc_challenger* c, const gl_t* e, size_t n);
gl_t orc_ch_challenge(orc_challenger* c);
gl2_t orc_ch_challenge_ext(orc_challenger* c);

/* ---------- K6: FRI fold of one layer (evaluation domain, bit-reversed order) ---------- */
/* in: m ext values in bit-reversed order of the coset shift*<w_m>; out: m/arity values of the folded
 * polynomial on shift^arity*<w_{m/arity}>, bit-reversed.  Equals plonky2's coefficient-space fold. */
void orc_fri_fold(const gl2_t* in, gl2_t* out, unsigned log_m, unsigned arity_bits, gl_t shift,
                  gl2_t beta);

/* ---------- synthetic STARK ---------- */
typedef struct {
  uint32_t log_n, n_cols, n_const, deg_pow, rate_bits, cap_height, num_queries, pow_bits,
      arity_bits, final_poly_bits;
  uint32_t air_id; /* 0: the synthetic AIR (stark.c), 1: Keccak-f[1600] (keccak_air.c), 2: logic (logic_air.c), 3: memory
                      (memory_air.c), 4: arithmetic (arithmetic_air.c), 5: byte packing
                      (byte_packing_air.c), 6: Keccak sponge (keccak_sponge_air.c),
                      7: multiplication (arithmetic_mul_air.c), 8: plonk (plonk_air.c); header word 14 of a proof */
  uint64_t pub[4]; /* the table's public inputs (AIR 8 binds them to its first row); zero for the other AIRs */
} orc_stark_cfg;
#define ORC_AIR_SYNTHETIC 0u
#define ORC_AIR_KECCAK_F 1u
#define ORC_KECCAK_COLS 2431u /* 2430 of the round + the lookup's filter column g (ctl.c), committed with the trace */
#define ORC_KECCAK_CONSTRAINTS 2826u
#define ORC_AIR_LOGIC 2u
#define ORC_LOGIC_COLS 524u
#define ORC_LOGIC_CONSTRAINTS 524u
#define ORC_AIR_MEMORY 3u
#define ORC_MEMORY_COLS 45u /* 44 of the log + the lookup's filter column g (ctl.c) */
#define ORC_MEMORY_CONSTRAINTS 60u
#define ORC_AIR_ARITHMETIC 4u
#define ORC_ARITHMETIC_COLS 309u
#define ORC_ARITHMETIC_CONSTRAINTS 294u
#define ORC_AIR_BYTE_PACKING 5u
#define ORC_BYTE_PACKING_COLS 299u
#define ORC_BYTE_PACKING_CONSTRAINTS 330u
#define ORC_AIR_KECCAK_SPONGE 6u
#define ORC_KECCAK_SPONGE_COLS 2414u
#define ORC_KECCAK_SPONGE_CONSTRAINTS 2587u
#define ORC_AIR_ARITHMETIC_MUL 7u
#define ORC_ARITHMETIC_MUL_COLS 1217u
#define ORC_ARITHMETIC_MUL_CONSTRAINTS 1218u
#define ORC_AIR_PLONK 8u
#define ORC_PLONK_COLS 135u
#define ORC_PLONK_CONSTS 85u

/* starky ConstraintConsumer: acc_j = acc_j * alpha_j + constraint, in list order; base field (the LDE coset)
 * and extension field (the verifier at zeta; the alphas stay in the base field). */
typedef struct {
  gl_t alpha[2], acc[2];
  gl_t z_last; /* x - g^-1 */
  gl_t l_first, l_last;
} orc_consumer;
static inline void orc_cons(orc_consumer* k, gl_t c) {
  k->acc[0] = gl_add(gl_mul(k->acc[0], k->alpha[0]), c);
  k->acc[1] = gl_add(gl_mul(k->acc[1], k->alpha[1]), c);
}
typedef struct {
  gl_t alpha[2]; gl2_t acc[2];
  gl2_t z_last, l_first, l_last;
} orc_consumer2;
static inline void orc_cons2(orc_consumer2* k, gl2_t c) {
  k->acc[0] = gl2_add(gl2_scale(k->acc[0], k->alpha[0]), c);
  k->acc[1] = gl2_add(gl2_scale(k->acc[1], k->alpha[1]), c);
}
/* keccak_air.c */
void orc_keccak_f(uint64_t lanes[25]);
void orc_keccak_trace(uint64_t seed, const uint64_t* inputs, unsigned log_n, gl_t* trace);
void orc_keccak_constraints_base(const gl_t* loc, const gl_t* nxt, orc_consumer* k);
void orc_keccak_constraints_ext(const gl2_t* loc, const gl2_t* nxt, orc_consumer2* k);
/* logic_air.c */
void orc_logic_trace(uint64_t seed, const uint64_t* inputs, unsigned log_n, gl_t* trace);
void orc_logic_constraints_base(const gl_t* loc, const gl_t* nxt, orc_consumer* k);
void orc_logic_constraints_ext(const gl2_t* loc, const gl2_t* nxt, orc_consumer2* k);
/* memory_air.c */
void orc_memory_trace(uint64_t seed, const uint64_t* inputs, unsigned log_n, gl_t* trace);
void orc_memory_constraints_base(const gl_t* loc, const gl_t* nxt, orc_consumer* k);
void orc_memory_constraints_ext(const gl2_t* loc, const gl2_t* nxt, orc_consumer2* k);
/* arithmetic_air.c */
void orc_arithmetic_trace(uint64_t seed, const uint64_t* inputs, unsigned log_n, gl_t* trace);
void orc_arithmetic_constraints_base(const gl_t* loc, const gl_t* nxt, orc_consumer* k);
void orc_arithmetic_constraints_ext(const gl2_t* loc, const gl2_t* nxt, orc_consumer2* k);
/* byte_packing_air.c */
void orc_byte_packing_trace(uint64_t seed, const uint64_t* inputs, unsigned log_n, gl_t* trace);
void orc_byte_packing_constraints_base(const gl_t* loc, const gl_t* nxt, orc_consumer* k);
void orc_byte_packing_constraints_ext(const gl2_t* loc, const gl2_t* nxt, orc_consumer2* k);
/* keccak_sponge_air.c */
size_t orc_keccak_sponge_rows(const uint8_t* msg, size_t len, uint64_t* rows, uint8_t* digest);
void orc_keccak_sponge_trace(uint64_t seed, const uint64_t* inputs, unsigned log_n, gl_t* trace);
/* the same; rows of a SEEDED table from row_limit on are padding rows */
void orc_keccak_sponge_trace_limit(uint64_t seed, const uint64_t* inputs, unsigned log_n, size_t row_limit, gl_t* trace);
void orc_keccak_sponge_constraints_base(const gl_t* loc, const gl_t* nxt, orc_consumer* k);
void orc_keccak_sponge_constraints_ext(const gl2_t* loc, const gl2_t* nxt, orc_consumer2* k);
/* plonk_air.c */
void orc_stark_public_inputs(uint64_t seed, gl_t out[4]);      /* hash of the list below */
void orc_stark_public_input_list(uint64_t seed, gl_t out[4]);
void orc_plonk_constants(uint64_t seed, unsigned log_n, unsigned pi_len, unsigned n_paths, unsigned depth, unsigned path_pi0,
                         unsigned leaf_len, gl_t* consts);
void orc_plonk_trace(uint64_t seed, const gl_t* pi, unsigned pi_len, unsigned n_paths, unsigned depth, unsigned path_pi0,
                     unsigned leaf_len, const gl_t* paths, const gl_t* consts, unsigned log_n, gl_t* trace);
void orc_plonk_aux_columns(const gl_t* trace_values, const gl_t* consts, unsigned log_n, const gl_t ctl[4], gl_t* aux);
void orc_plonk_constraints_base(const gl_t* cst, const gl_t* loc, const gl_t* aux, const gl_t* aux_nxt, const gl_t ctl[4],
                                const gl_t pub[4], gl_t x, orc_consumer* k);
void orc_plonk_constraints_ext(const gl2_t* cst, const gl2_t* loc, const gl2_t* aux, const gl2_t* aux_nxt, const gl_t ctl[4],
                               const gl_t pub[4], gl2_t x, orc_consumer2* k);
/* arithmetic_mul_air.c */
void orc_arithmetic_mul_trace(uint64_t seed, const uint64_t* inputs, unsigned log_n, gl_t* trace);
void orc_arithmetic_mul_constraints_base(const gl_t* loc, const gl_t* nxt, orc_consumer* k);
void orc_arithmetic_mul_constraints_ext(const gl2_t* loc, const gl2_t* nxt, orc_consumer2* k);

/* ctl.c: the cross-table lookups (auxiliary columns) of the tables with a real AIR */
uint32_t orc_ctl_n_aux(uint32_t air_id, uint32_t n_cols);
void orc_ctl_aux_columns(uint32_t air_id, const gl_t* trace_values, unsigned log_n, const gl_t ctl[4], gl_t* aux);
/* the filter column of a LOOKED table's trace (Keccak-f: column 2430, exposed[p] per permutation; memory: column 44,
 * exposed[i] per row), set before the trace is committed; NULL: nothing is exposed */
void orc_ctl_set_filter(uint32_t air_id, gl_t* trace_values, unsigned log_n, const uint8_t* exposed, size_t n_exposed);
void orc_ctl_constraints_base(uint32_t air_id, const gl_t* loc, const gl_t* aux, const gl_t* aux_nxt, const gl_t ctl[4],
                              orc_consumer* k);
void orc_ctl_constraints_ext(uint32_t air_id, const gl2_t* loc, const gl2_t* aux, const gl2_t* aux_nxt, const gl_t ctl[4],
                             orc_consumer2* k);

uint32_t orc_cfg_n_aux(const orc_stark_cfg* c);
uint32_t orc_cfg_n_quot(const orc_stark_cfg* c);
uint32_t orc_cfg_n_layers(const orc_stark_cfg* c);
size_t orc_proof_words(const orc_stark_cfg* c);

/* deterministic witness: fills n_cols columns of 2^log_n rows (column-major) satisfying the AIR.
 * consts: n_const columns (NULL when n_const == 0). */
void orc_synth_constants(uint64_t seed, unsigned log_n, size_t n_const, gl_t* consts);
void orc_synth_trace(uint64_t seed, const orc_stark_cfg* c, const gl_t* consts, gl_t* trace);

/* preprocessed constants commitment (ProverState analog, prover_state.rs:80-100) */
/* K5: quotient values on the LDE coset in natural order; LDE matrices column-major with stride n << rate_bits
 * (const_lde may be NULL when n_const == 0).  qv: [2][n << rate_bits]. */
void orc_quotient_values(const orc_stark_cfg* c, const gl_t* const_lde, const gl_t* trace_lde, const gl_t* aux_lde,
                         const gl_t ctl[4], gl_t alpha0, gl_t alpha1, gl_t* qv);

typedef struct orc_committed orc_committed;
orc_committed* orc_commit_values(const gl_t* values, unsigned log_n, size_t n_cols, unsigned rate_bits,
                                 unsigned cap_h);
orc_committed* orc_commit_coeffs(const gl_t* coeffs, unsigned log_n, size_t n_cols, unsigned rate_bits,
                                 unsigned cap_h);
const gl_t* orc_committed_cap(const orc_committed* c);
const gl_t* orc_committed_lde(const orc_committed* c);
const gl_t* orc_committed_coeffs(const orc_committed* c);
const gl_t* orc_committed_digests(const orc_committed* c);
void orc_committed_free(orc_committed* c);

/* Prove one table.  `ch` is the running transcript (trace cap NOT yet observed by this call: the
 * caller observes caps first, plonky2_evm prover order).  ctl[4] = beta0,gamma0,beta1,gamma1.
 * Returns 0 on success. proof_out has orc_proof_words(cfg) words. */
int orc_stark_prove(const orc_stark_cfg* cfg, const orc_committed* consts, const orc_committed* trace,
                    const gl_t* trace_values, const gl_t ctl[4], orc_challenger* ch, gl_t* proof_out);
/* Verify; the caller must have driven `ch` identically (caps observed etc.). 0 = accept. */
int orc_stark_verify(const orc_stark_cfg* cfg, const gl_t* const_cap, const gl_t ctl[4],
                     orc_challenger* ch, const gl_t* proof);
void orc_proof_first_query_path(const orc_stark_cfg* cf, const gl_t* proof, gl_t leaf[4], gl_t cap_entry[4], gl_t* path);
void orc_proof_first_query_row(const orc_stark_cfg* cf, const gl_t* proof, gl_t* row); /* the opened trace row itself (n_cols words) */
void orc_proof_digest(const orc_stark_cfg* cfg, const gl_t* proof, gl_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
