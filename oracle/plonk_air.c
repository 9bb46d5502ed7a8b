/* oracle/plonk_air.c -- AIR 8: a PLONK-shaped circuit as a table (CPU restatement).  TEST INFRASTRUCTURE ONLY; "parity
 * unpinned" by the reference (see gl.h): upstream proves every recursion circuit -- the per-table shrink chains of
 * generate_txn_proof (plonky_block_proof_gen/src/proof_gen.rs:44-52), prove_aggregation (:66-75) and prove_block
 * (:97-103) -- with plonky2's CircuitData::prove, which is not under /root/reference.  What is restated here is the
 * SHAPE of that proof system from its public description (PLONK with plonky2's partial-products form of the
 * permutation argument; CircuitConfig::standard_recursion_config: 135 wires, 80 routed, degree factor 8
 * [UPSTREAM-UNVERIFIED]): gates selected by preprocessed constants, public inputs bound to the first row, copy
 * constraints.  NOT upstream's gate set, NOT a verifier circuit.
 *
 * Wires: 20 slots (a, b, c, d) = columns 4s..4s+3 (routed), 11 S-box units (x, x^2, x^4, x^6, x^7) = columns 80+5i...
 * Constants (84): 0 q_arith, 1 q_sbox, 2 c0, 3 c1, 4 + j sigma_j.  Copy permutation: wire (j, i) is the field element
 * 7^j w^i; sigma maps every wire to the next one of its equivalence class.  The circuit: rows in groups of four,
 *   4g   arith, free inputs     4g+1  arith on the outputs of 4g     4g+2  S-box of the first 11 outputs of 4g+1
 *   4g+3 arith whose first 11 `a` inputs are the S-box outputs;  group 0 = the public-input row and three no-ops;
 * c_j of row 4 (j < 4) is the public input w_j of row 0.  Equivalence classes are listed below as explicit sets; the
 * product (csrc/air.hpp, plonk::sigma_of) computes "the next member" in closed form. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

enum { PW = 135, PK = 84, ROUTED = 80, SLOTS = 20, SBOX = 11, SB0 = 80 };

static inline uint64_t smix(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static inline gl_t rnd(uint64_t seed, uint64_t col, uint64_t row) { return gl_canon(smix(seed ^ (col << 32) ^ row)); }

void orc_stark_public_inputs(uint64_t seed, gl_t out[4]) { /* of a lone table proof: from its seed */
  for (uint64_t j = 0; j < 4; j++) out[j] = gl_canon(smix(seed ^ ((0x50 + j) << 32)));
}

typedef struct { uint32_t col, row; } wire_t;
/* ties the wires of one class into a cycle: each member's sigma is the next member, the last one's the first */
static void tie(gl_t* consts, size_t N, const gl_t* kpow, const gl_t* wpow, const wire_t* m, int n) {
  for (int i = 0; i < n; i++) {
    const wire_t to = m[(i + 1) % n];
    consts[(size_t)(4 + m[i].col) * N + m[i].row] = gl_mul(kpow[to.col], wpow[to.row]);
  }
}
void orc_plonk_constants(uint64_t seed, unsigned log_n, gl_t* consts) {
  const size_t N = (size_t)1 << log_n;
  gl_t *kpow = (gl_t*)malloc(ROUTED * sizeof(gl_t)), *wpow = (gl_t*)malloc(N * sizeof(gl_t));
  kpow[0] = 1;
  for (int j = 1; j < ROUTED; j++) kpow[j] = gl_mul(kpow[j - 1], 7);
  wpow[0] = 1;
  for (size_t i = 1; i < N; i++) wpow[i] = gl_mul(wpow[i - 1], gl_root(log_n));
  for (size_t i = 0; i < N; i++) {
    const int gate_row = i >= 4;
    consts[0 * N + i] = gate_row && (i % 4 != 2);
    consts[1 * N + i] = gate_row && (i % 4 == 2);
    consts[2 * N + i] = rnd(seed ^ 0xC0115700C0115700ULL, 2, i);
    consts[3 * N + i] = rnd(seed ^ 0xC0115700C0115700ULL, 3, i);
    for (int j = 0; j < ROUTED; j++) consts[(size_t)(4 + j) * N + i] = gl_mul(kpow[j], wpow[i]); /* untied: itself */
  }
  for (size_t g = 1; g < N / 4; g++) {
    const uint32_t r = (uint32_t)(4 * g);
    for (uint32_t s = 0; s < SLOTS; s++) {
      /* d_s(4g) is the `a` input of slot s and the `b` input of slot s - 1 in the next row */
      const wire_t dab[3] = {{4 * s + 3, r}, {4 * s, r + 1}, {4 * ((s + SLOTS - 1) % SLOTS) + 1, r + 1}};
      tie(consts, N, kpow, wpow, dab, 3);
      if (g == 1 && s < 4) { /* c_s of the first computing rows is public input s */
        const wire_t cc[3] = {{4 * s + 2, r}, {4 * s + 2, r + 1}, {s, 0}};
        tie(consts, N, kpow, wpow, cc, 3);
      } else {
        const wire_t cc[2] = {{4 * s + 2, r}, {4 * s + 2, r + 1}};
        tie(consts, N, kpow, wpow, cc, 2);
      }
      if (s < SBOX) {
        const wire_t in[2] = {{4 * s + 3, r + 1}, {4 * s, r + 2}};  /* S-box input = output s of the row above */
        const wire_t out[2] = {{4 * s + 3, r + 2}, {4 * s, r + 3}}; /* S-box output = `a` input s of the row below */
        tie(consts, N, kpow, wpow, in, 2);
        tie(consts, N, kpow, wpow, out, 2);
      }
    }
  }
  free(kpow); free(wpow);
}

/* Witness: row by row.  consts: the circuit's constants (c0, c1 are read). */
void orc_plonk_trace(uint64_t seed, const gl_t pub[4], const gl_t* consts, unsigned log_n, gl_t* t) {
  const size_t N = (size_t)1 << log_n;
#define W(col, row) t[(size_t)(col) * N + (row)]
  for (size_t i = 0; i < N; i++) /* every wire starts free; the rows below overwrite what the circuit computes */
    for (int c = 0; c < PW; c++) W(c, i) = rnd(seed, c, i);
  for (int j = 0; j < 4; j++) W(j, 0) = pub[j];
  for (size_t i = 4; i < N; i++) {
    const gl_t c0 = consts[2 * N + i], c1 = consts[3 * N + i];
    const int p = (int)(i % 4);
    if (p == 1)
      for (int s = 0; s < SLOTS; s++) {
        W(4 * s, i) = W(4 * s + 3, i - 1);
        W(4 * s + 1, i) = W(4 * ((s + 1) % SLOTS) + 3, i - 1);
        W(4 * s + 2, i) = W(4 * s + 2, i - 1);
      }
    if (p == 0 && i == 4)
      for (int s = 0; s < 4; s++) W(4 * s + 2, i) = pub[s];
    if (p == 3)
      for (int s = 0; s < SBOX; s++) W(4 * s, i) = W(4 * s + 3, i - 1);
    if (p == 2) {
      for (int s = 0; s < SBOX; s++) {
        const gl_t x = W(4 * s + 3, i - 1), x2 = gl_mul(x, x), x4 = gl_mul(x2, x2), x6 = gl_mul(x4, x2), x7 = gl_mul(x6, x);
        W(4 * s, i) = x; W(4 * s + 3, i) = x7;
        W(SB0 + 5 * s, i) = x; W(SB0 + 5 * s + 1, i) = x2; W(SB0 + 5 * s + 2, i) = x4; W(SB0 + 5 * s + 3, i) = x6;
        W(SB0 + 5 * s + 4, i) = x7;
      }
    } else {
      for (int s = 0; s < SLOTS; s++)
        W(4 * s + 3, i) = gl_add(gl_mul(c0, gl_mul(W(4 * s, i), W(4 * s + 1, i))), gl_mul(c1, W(4 * s + 2, i)));
    }
  }
#undef W
}

/* The copy-constraint columns.  Per challenge set: Z (column 10 c) and nine partial products (10 c + k).  With
 * ratio_k(i) = prod_{j in chunk k} (w_j + beta 7^j w^i + gamma) / (w_j + beta sigma_j + gamma):
 *   Z(i) = prod_{i' >= i} prod_k ratio_k(i')       (Z(0) = 1 when the copy constraints hold)
 *   partial product k of row i = Z(i + 1) ratio_1(i) ... ratio_k(i),  rows wrapping (Z(N) = Z(0)). */
void orc_plonk_aux_columns(const gl_t* tv, const gl_t* consts, unsigned log_n, const gl_t ctl[4], gl_t* aux) {
  const size_t N = (size_t)1 << log_n;
  gl_t* ratio = (gl_t*)malloc(10 * N * sizeof(gl_t));
  for (int c = 0; c < 2; c++) {
    const gl_t beta = ctl[2 * c], gamma = ctl[2 * c + 1];
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < N; i++) {
      const gl_t x = gl_pow(gl_root(log_n), i);
      gl_t kj = 1;
      for (int k = 0; k < 10; k++) {
        gl_t num = 1, den = 1;
        for (int j = 8 * k; j < 8 * k + 8; j++) {
          const gl_t w = tv[(size_t)j * N + i];
          num = gl_mul(num, gl_add(gl_add(w, gl_mul(beta, gl_mul(kj, x))), gamma));
          den = gl_mul(den, gl_add(gl_add(w, gl_mul(beta, consts[(size_t)(4 + j) * N + i])), gamma));
          kj = gl_mul(kj, 7);
        }
        ratio[(size_t)k * N + i] = gl_mul(num, gl_inv(den));
      }
    }
    gl_t* Z = aux + (size_t)(10 * c) * N;
    gl_t run = 1;
    for (size_t i = N; i-- > 0;) {
      for (int k = 0; k < 10; k++) run = gl_mul(run, ratio[(size_t)k * N + i]);
      Z[i] = run;
    }
    for (size_t i = 0; i < N; i++) {
      gl_t p = Z[(i + 1) % N];
      for (int k = 1; k <= 9; k++) {
        p = gl_mul(p, ratio[(size_t)(k - 1) * N + i]);
        aux[(size_t)(10 * c + k) * N + i] = p;
      }
    }
  }
  free(ratio);
}

/* ---- constraints, base field ---- */
#define FT gl_t
#define FK(c) ((gl_t)(c))
#define FADD gl_add
#define FSUB gl_sub
#define FMUL gl_mul
#define FSCALE(a, s) gl_mul(a, s)
#define FNAME(n) pb_##n
#define CONS_T orc_consumer
#define CONS_ALL(k, c) orc_cons(k, c)
#define CONS_FIRST(k, c) orc_cons(k, gl_mul(c, (k)->l_first))
#include "plonk_air_body.inc"
#undef FT
#undef FK
#undef FADD
#undef FSUB
#undef FMUL
#undef FSCALE
#undef FNAME
#undef CONS_T
#undef CONS_ALL
#undef CONS_FIRST
void orc_plonk_constraints_base(const gl_t* cst, const gl_t* loc, const gl_t* aux, const gl_t* aux_nxt, const gl_t ctl[4],
                                const gl_t pub[4], gl_t x, orc_consumer* k) {
  pb_plonk_constraints(cst, loc, aux, aux_nxt, ctl, pub, x, k);
}
/* ---- the same over the extension ---- */
#define FT gl2_t
#define FK(c) gl2_from((gl_t)(c))
#define FADD gl2_add
#define FSUB gl2_sub
#define FMUL gl2_mul
#define FSCALE(a, s) gl2_scale(a, s)
#define FNAME(n) pe_##n
#define CONS_T orc_consumer2
#define CONS_ALL(k, c) orc_cons2(k, c)
#define CONS_FIRST(k, c) orc_cons2(k, gl2_mul(c, (k)->l_first))
#include "plonk_air_body.inc"
void orc_plonk_constraints_ext(const gl2_t* cst, const gl2_t* loc, const gl2_t* aux, const gl2_t* aux_nxt, const gl_t ctl[4],
                               const gl_t pub[4], gl2_t x, orc_consumer2* k) {
  pe_plonk_constraints(cst, loc, aux, aux_nxt, ctl, pub, x, k);
}
