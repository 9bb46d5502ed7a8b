/* oracle/ctl.c -- cross-table lookups (CPU restatement).  TEST INFRASTRUCTURE ONLY; "parity unpinned" by the
 * reference (see gl.h): upstream proves the seven tables of a transaction as ONE statement (one prove_root call,
 * plonky_block_proof_gen/src/proof_gen.rs:44-52; tables prover_state.rs:85-93) and ties them with plonky2_evm's
 * cross-table lookups, which are not under /root/reference.  The lookups here are this repository's own, written for
 * its own column layouts (AIRS.md section 3):
 *
 *   keccak_sponge -> keccak_f: every row of the sponge table that absorbs a block claims "the permutation of (xored
 *   rate, capacity) is (updated state)"; the Keccak-f table exposes (input, output) of the permutations it is asked for.
 *
 *   keccak_sponge -> logic: every row of the sponge table that absorbs a block claims "the XOR of limbs 8 m .. 8 m + 7 of
 *   the rate with the block is an operation of the logic table", m < 5 (136 bytes = 34 limbs: the fifth operation has two
 *   limbs and six zeros); the logic table exposes the operations it is asked for (filter: trace column 523).
 *
 *   byte_packing -> memory: every row of the byte-packing table that moves a word claims "the memory operation
 *   (is_read, address, timestamp) carries this 256-bit value"; the memory table exposes the operations it is asked for.
 *
 * A lookup is a pair of filtered running products z[i] = prod_{i' >= i} factor[i'], factor = (gamma + sum_j beta^j
 * tuple_j) on the rows that take part and 1 on the others; the statement holds when the first-row values agree, for
 * both challenge sets (beta_0, gamma_0), (beta_1, gamma_1).  Auxiliary columns per table:
 *   synthetic  n_cols / 8   unfiltered products over trace columns 8k, 8k + 1 (a load placeholder; stark.c)
 *   keccak_f   4            h_0, h_1 (the permutation's 50 input limbs compressed by beta_c, the same value on all its
 *                           rows), z_0, z_1
 *   sponge     12           z_0, z_1 (-> keccak_f); then for m < 5 and challenge set c column 2 + 2 m + c (-> logic)
 *   logic      2            z_0, z_1
 *   byte_packing 2          z_0, z_1
 *   memory     2            z_0, z_1
 * The FILTERS of the two looked tables -- g = 1 on the last-round row of an exposed permutation (Keccak-f, trace column
 * 2430), on an exposed operation (memory, trace column 44) -- are columns of the TRACE: set by orc_ctl_set_filter before
 * the trace is committed, i.e. before the challenges are drawn (as auxiliary columns, committed after the challenges,
 * they left the prover free to pick the exposed subset knowing beta and gamma: ADVICE r4).
 *   others     1            the constant 1
 * This file computes the columns by their meaning (if / else per row, explicit powers) and states the constraints as
 * polynomials; the product (csrc/air.hpp, namespace ctl) evaluates the same polynomials by Horner's rule. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

enum { KCOL_STEP = 0, KCOL_A = 24, KCOL_APP = 2314, KCOL_APPP = 2428, KCOL_G = 2430 };
enum { SCOL_FULL = 0, SCOL_FINAL = 1, SCOL_CAP = 2314, SCOL_XORED = 2330, SCOL_UPDATED = 2364 };
enum { PCOL_READ = 0, PCOL_LEN = 1, PCOL_VAL = 289, PCOL_ADDR = 297, PCOL_TS = 298 };
enum { MCOL_READ = 0, MCOL_ADDR = 1, MCOL_TS = 2, MCOL_VAL = 3, MCOL_G = 44 };
enum { SCOL_BLOCK = 138, SCOL_RATE = 1226 };                       /* the sponge table's bit columns (block as absorbed, rate before) */
enum { LCOL_OP = 0, LCOL_IN0 = 3, LCOL_IN1 = 259, LCOL_RES = 515, LCOL_G = 523 };

uint32_t orc_ctl_n_aux(uint32_t air_id, uint32_t n_cols) {
  return air_id == ORC_AIR_SYNTHETIC ? n_cols / 8
         : air_id == ORC_AIR_KECCAK_F ? 4
         : air_id == ORC_AIR_KECCAK_SPONGE ? 12 /* z_0 z_1 (-> keccak_f), then five logic operations x two challenge sets */
         : air_id == ORC_AIR_BYTE_PACKING ? 2
         : air_id == ORC_AIR_MEMORY ? 2
         : air_id == ORC_AIR_LOGIC ? 2
         : air_id == ORC_AIR_PLONK ? 20 /* Z + nine partial products per challenge set (plonk_air.c) */
                                   : 1;
}

/* The filter column of a LOOKED table's trace tv ([n_cols][N], column-major), before the trace is committed.
 * Keccak-f table: exposed[p] != 0 when permutation p is asked for; memory table: exposed[i] != 0 when the operation in
 * row i is asked for; n_exposed entries, NULL: nothing is exposed. */
void orc_ctl_set_filter(uint32_t air_id, gl_t* tv, unsigned log_n, const uint8_t* exposed, size_t n_exposed) {
  const size_t N = (size_t)1 << log_n;
  if (air_id == ORC_AIR_KECCAK_F)
    for (size_t i = 0; i < N; i++) tv[(size_t)KCOL_G * N + i] = (i % 24 == 23 && exposed && i / 24 < n_exposed && exposed[i / 24]) ? 1 : 0;
  else if (air_id == ORC_AIR_MEMORY)
    for (size_t i = 0; i < N; i++) tv[(size_t)MCOL_G * N + i] = (exposed && i < n_exposed && exposed[i]) ? 1 : 0;
  else if (air_id == ORC_AIR_LOGIC) /* exposed[i]: the operation in row i is asked for */
    for (size_t i = 0; i < N; i++) tv[(size_t)LCOL_G * N + i] = (exposed && i < n_exposed && exposed[i]) ? 1 : 0;
}

/* keccak_sponge -> logic.  The 27-word tuple of a logic operation: the three flags, the eight 32-bit limbs of either
 * input, the eight of the result.  As the sponge table sends it for the m-th group of eight rate limbs of row i (an XOR
 * of the rate with the block; limbs past the 34th do not exist: zero), and as the logic table offers it for row i. */
static void sponge_logic_tuple(const gl_t* tv, size_t N, size_t i, int m, gl_t t[27]) {
  t[0] = 0; t[1] = 0; t[2] = 1;
  for (int j = 0; j < 8; j++) {
    const int l = 8 * m + j;
    gl_t rate = 0, block = 0;
    if (l < 34)
      for (int z = 0; z < 32; z++) {
        rate += tv[(size_t)(SCOL_RATE + 32 * l + z) * N + i] << z;
        block += tv[(size_t)(SCOL_BLOCK + 32 * l + z) * N + i] << z;
      }
    t[3 + j] = rate; t[11 + j] = block;
    t[19 + j] = l < 34 ? tv[(size_t)(SCOL_XORED + l) * N + i] : 0;
  }
}
static void logic_tuple(const gl_t* tv, size_t N, size_t i, gl_t t[27]) {
  for (int j = 0; j < 3; j++) t[j] = tv[(size_t)(LCOL_OP + j) * N + i];
  for (int j = 0; j < 8; j++) {
    gl_t a = 0, b = 0;
    for (int z = 0; z < 32; z++) {
      a += tv[(size_t)(LCOL_IN0 + 32 * j + z) * N + i] << z;
      b += tv[(size_t)(LCOL_IN1 + 32 * j + z) * N + i] << z;
    }
    t[3 + j] = a; t[11 + j] = b; t[19 + j] = tv[(size_t)(LCOL_RES + j) * N + i];
  }
}
static gl_t compress27(const gl_t t[27], gl_t beta) {
  gl_t acc = 0, pw = 1;
  for (int j = 0; j < 27; j++) { acc = gl_add(acc, gl_mul(pw, gl_canon(t[j]))); pw = gl_mul(pw, beta); }
  return acc;
}

/* The auxiliary columns of a table with a real AIR, from its trace values tv ([n_cols][N], column-major). */
void orc_ctl_aux_columns(uint32_t air_id, const gl_t* tv, unsigned log_n, const gl_t ctl[4], gl_t* aux) {
  const size_t N = (size_t)1 << log_n;
  if (air_id == ORC_AIR_KECCAK_F) {
    const gl_t* g = tv + (size_t)KCOL_G * N;
    gl_t *h[2] = {aux, aux + N}, *z[2] = {aux + 2 * N, aux + 3 * N};
    for (int c = 0; c < 2; c++) {
      const gl_t beta = ctl[2 * c], gamma = ctl[2 * c + 1];
      gl_t pw[100];
      pw[0] = 1;
      for (int j = 1; j < 100; j++) pw[j] = gl_mul(pw[j - 1], beta);
#pragma omp parallel for schedule(static)
      for (size_t p = 0; p < (N + 23) / 24; p++) { /* the compressed input, on every row of the permutation */
        gl_t acc = 0;
        for (int j = 0; j < 50; j++) acc = gl_add(acc, gl_mul(pw[j], tv[(size_t)(KCOL_A + j) * N + 24 * p]));
        for (size_t i = 24 * p; i < 24 * p + 24 && i < N; i++) h[c][i] = acc;
      }
      gl_t run = 1;
      for (size_t i = N; i-- > 0;) {
        if (g[i]) {
          gl_t tuple = h[c][i];
          for (int j = 0; j < 50; j++) {
            const size_t col = j < 2 ? KCOL_APPP + j : KCOL_APP + j;
            tuple = gl_add(tuple, gl_mul(pw[50 + j], tv[col * N + i]));
          }
          run = gl_mul(run, gl_add(gamma, tuple));
        }
        z[c][i] = run;
      }
    }
    return;
  }
  if (air_id == ORC_AIR_KECCAK_SPONGE) {
    for (int c = 0; c < 2; c++) {
      const gl_t beta = ctl[2 * c], gamma = ctl[2 * c + 1];
      gl_t pw[100], *z = aux + (size_t)c * N, run = 1;
      pw[0] = 1;
      for (int j = 1; j < 100; j++) pw[j] = gl_mul(pw[j - 1], beta);
      for (size_t i = N; i-- > 0;) {
        if (tv[(size_t)SCOL_FULL * N + i] || tv[(size_t)SCOL_FINAL * N + i]) {
          gl_t tuple = 0;
          for (int j = 0; j < 34; j++) tuple = gl_add(tuple, gl_mul(pw[j], tv[(size_t)(SCOL_XORED + j) * N + i]));
          for (int j = 0; j < 16; j++) tuple = gl_add(tuple, gl_mul(pw[34 + j], tv[(size_t)(SCOL_CAP + j) * N + i]));
          for (int j = 0; j < 50; j++) tuple = gl_add(tuple, gl_mul(pw[50 + j], tv[(size_t)(SCOL_UPDATED + j) * N + i]));
          run = gl_mul(run, gl_add(gamma, tuple));
        }
        z[i] = run;
      }
    }
    for (int m = 0; m < 5; m++) /* keccak_sponge -> logic: column 2 + 2 m + c */
      for (int c = 0; c < 2; c++) {
        gl_t *z = aux + (size_t)(2 + 2 * m + c) * N, run = 1;
        for (size_t i = N; i-- > 0;) {
          if (tv[(size_t)SCOL_FULL * N + i] || tv[(size_t)SCOL_FINAL * N + i]) {
            gl_t t[27];
            sponge_logic_tuple(tv, N, i, m, t);
            run = gl_mul(run, gl_add(ctl[2 * c + 1], compress27(t, ctl[2 * c])));
          }
          z[i] = run;
        }
      }
    return;
  }
  if (air_id == ORC_AIR_LOGIC) { /* an exposed row offers its operation */
    for (int c = 0; c < 2; c++) {
      gl_t *z = aux + (size_t)c * N, run = 1;
      for (size_t i = N; i-- > 0;) {
        if (tv[(size_t)LCOL_G * N + i]) {
          gl_t t[27];
          logic_tuple(tv, N, i, t);
          run = gl_mul(run, gl_add(ctl[2 * c + 1], compress27(t, ctl[2 * c])));
        }
        z[i] = run;
      }
    }
    return;
  }
  if (air_id == ORC_AIR_BYTE_PACKING) { /* a row with a length sends (is_read, address, timestamp, value limbs) */
    for (int c = 0; c < 2; c++) {
      const gl_t beta = ctl[2 * c], gamma = ctl[2 * c + 1];
      gl_t *z = aux + (size_t)c * N, run = 1;
      for (size_t i = N; i-- > 0;) {
        int has_len = 0;
        for (int j = 0; j < 32; j++) has_len |= tv[(size_t)(PCOL_LEN + j) * N + i] != 0;
        if (has_len) {
          gl_t tuple = tv[(size_t)PCOL_READ * N + i], pw = beta;
          tuple = gl_add(tuple, gl_mul(pw, tv[(size_t)PCOL_ADDR * N + i])); pw = gl_mul(pw, beta);
          tuple = gl_add(tuple, gl_mul(pw, tv[(size_t)PCOL_TS * N + i])); pw = gl_mul(pw, beta);
          for (int k = 0; k < 8; k++) { tuple = gl_add(tuple, gl_mul(pw, tv[(size_t)(PCOL_VAL + k) * N + i])); pw = gl_mul(pw, beta); }
          run = gl_mul(run, gl_add(gamma, tuple));
        }
        z[i] = run;
      }
    }
    return;
  }
  if (air_id == ORC_AIR_MEMORY) { /* an exposed row offers (is_read, address, timestamp, value limbs) */
    const gl_t* g = tv + (size_t)MCOL_G * N;
    for (int c = 0; c < 2; c++) {
      const gl_t beta = ctl[2 * c], gamma = ctl[2 * c + 1];
      gl_t *z = aux + (size_t)c * N, run = 1;
      for (size_t i = N; i-- > 0;) {
        if (g[i]) {
          gl_t tuple = tv[(size_t)MCOL_READ * N + i], pw = beta;
          tuple = gl_add(tuple, gl_mul(pw, tv[(size_t)MCOL_ADDR * N + i])); pw = gl_mul(pw, beta);
          tuple = gl_add(tuple, gl_mul(pw, tv[(size_t)MCOL_TS * N + i])); pw = gl_mul(pw, beta);
          for (int k = 0; k < 8; k++) { tuple = gl_add(tuple, gl_mul(pw, tv[(size_t)(MCOL_VAL + k) * N + i])); pw = gl_mul(pw, beta); }
          run = gl_mul(run, gl_add(gamma, tuple));
        }
        z[i] = run;
      }
    }
    return;
  }
  for (size_t i = 0; i < N; i++) aux[i] = 1;
}

/* ---- constraints, base field ---- */
#define FT gl_t
#define FK(c) ((gl_t)(c))
#define FADD gl_add
#define FSUB gl_sub
#define FMUL gl_mul
#define FSCALE(a, s) gl_mul(a, s)
#define FNAME(n) cb_##n
#define CONS_T orc_consumer
#define CONS_ALL(k, c) orc_cons(k, c)
#define CONS_TRANS(k, c) orc_cons(k, gl_mul(c, (k)->z_last))
#define CONS_LAST(k, c) orc_cons(k, gl_mul(c, (k)->l_last))
#include "ctl_body.inc"
#undef FT
#undef FK
#undef FADD
#undef FSUB
#undef FMUL
#undef FSCALE
#undef FNAME
#undef CONS_T
#undef CONS_ALL
#undef CONS_TRANS
#undef CONS_LAST
void orc_ctl_constraints_base(uint32_t air_id, const gl_t* loc, const gl_t* aux, const gl_t* aux_nxt, const gl_t ctl[4],
                              orc_consumer* k) {
  cb_ctl_constraints(air_id, loc, aux, aux_nxt, ctl, k);
}
/* ---- the same over the extension ---- */
#define FT gl2_t
#define FK(c) gl2_from((gl_t)(c))
#define FADD gl2_add
#define FSUB gl2_sub
#define FMUL gl2_mul
#define FSCALE(a, s) gl2_scale(a, s)
#define FNAME(n) ce_##n
#define CONS_T orc_consumer2
#define CONS_ALL(k, c) orc_cons2(k, c)
#define CONS_TRANS(k, c) orc_cons2(k, gl2_mul(c, (k)->z_last))
#define CONS_LAST(k, c) orc_cons2(k, gl2_mul(c, (k)->l_last))
#include "ctl_body.inc"
void orc_ctl_constraints_ext(uint32_t air_id, const gl2_t* loc, const gl2_t* aux, const gl2_t* aux_nxt, const gl_t ctl[4],
                             orc_consumer2* k) {
  ce_ctl_constraints(air_id, loc, aux, aux_nxt, ctl, k);
}
