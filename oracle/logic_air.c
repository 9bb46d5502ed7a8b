/* oracle/logic_air.c -- AIR 2: one bitwise operation (AND / OR / XOR) on two 256-bit words per trace row, 524 columns
 * (the last one is the filter of the lookup keccak_sponge -> logic, ctl.c).
 * TEST INFRASTRUCTURE ONLY; "parity unpinned" by the reference (see gl.h): the reference proves its logic table
 * through the out-of-tree plonky2_evm (call site plonky_block_proof_gen/src/proof_gen.rs:44-52, table list
 * prover_state.rs:85-93 "logic", size range constants.rs:14); nothing under /root/reference shows its columns.
 * Written from the definition of the three operations; the tests check the trace against Python's big integers.
 *
 * Column map (shared with the product by specification):
 *   0..2 is_and, is_or, is_xor | 3..258 bits of operand 0 | 259..514 bits of operand 1 | 515..522 result limbs (u32) */
#include "oracle.h"
#include <string.h>

enum { LG_OP = 0, LG_IN0 = 3, LG_IN1 = 259, LG_RES = 515 };

static inline uint64_t smix(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* Witness: n = 2^log_n rows x 524 columns, column-major.  inputs: [n][9] = operation code (0 none, 1 and, 2 or,
 * 3 xor), operand 0 (four u64, least significant first), operand 1; or NULL, then row r draws
 * code = smix(seed ^ (0xFF << 32) ^ r) & 3 and word w of operand j = smix(seed ^ ((1 + 4j + w) << 32) ^ r). */
void orc_logic_trace(uint64_t seed, const uint64_t* inputs, unsigned log_n, gl_t* t) {
  const size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < n; r++) {
    uint64_t x[4], y[4], res[4];
    const unsigned op = (unsigned)((inputs ? inputs[r * 9] : smix(seed ^ (0xFFULL << 32) ^ r)) & 3);
    for (int w = 0; w < 4; w++) {
      x[w] = inputs ? inputs[r * 9 + 1 + w] : smix(seed ^ ((uint64_t)(1 + w) << 32) ^ r);
      y[w] = inputs ? inputs[r * 9 + 5 + w] : smix(seed ^ ((uint64_t)(5 + w) << 32) ^ r);
      res[w] = op == 1 ? (x[w] & y[w]) : op == 2 ? (x[w] | y[w]) : op == 3 ? (x[w] ^ y[w]) : 0;
    }
#define PUT(col, v) t[(size_t)(col) * n + r] = (gl_t)(v)
    PUT(LG_OP, op == 1);
    PUT(LG_OP + 1, op == 2);
    PUT(LG_OP + 2, op == 3);
    for (int i = 0; i < 256; i++) {
      PUT(LG_IN0 + i, (x[i / 64] >> (i % 64)) & 1);
      PUT(LG_IN1 + i, (y[i / 64] >> (i % 64)) & 1);
    }
    for (int limb = 0; limb < 8; limb++) PUT(LG_RES + limb, (res[limb / 2] >> (32 * (limb % 2))) & 0xFFFFFFFFULL);
    PUT(523, 0); /* the lookup's filter (ctl.c: keccak_sponge -> logic), set by orc_ctl_set_filter */
#undef PUT
  }
}

/* ---- constraints, base field (the quotient on the LDE coset) ---- */
#define FT gl_t
#define FK(c) ((gl_t)(c))
#define FADD gl_add
#define FSUB gl_sub
#define FMUL gl_mul
#define FNAME(n) lb_##n
#define CONS_T orc_consumer
#define CONS_ALL(k, c) orc_cons(k, c)
#include "logic_air_body.inc"
#undef FT
#undef FK
#undef FADD
#undef FSUB
#undef FMUL
#undef FNAME
#undef CONS_T
#undef CONS_ALL
void orc_logic_constraints_base(const gl_t* loc, const gl_t* nxt, orc_consumer* k) { (void)nxt; lb_logic_constraints(loc, k); }

/* ---- the same over the extension (the verifier's check at zeta) ---- */
#define FT gl2_t
#define FK(c) gl2_from((gl_t)(c))
#define FADD gl2_add
#define FSUB gl2_sub
#define FMUL gl2_mul
#define FNAME(n) le_##n
#define CONS_T orc_consumer2
#define CONS_ALL(k, c) orc_cons2(k, c)
#include "logic_air_body.inc"
void orc_logic_constraints_ext(const gl2_t* loc, const gl2_t* nxt, orc_consumer2* k) { (void)nxt; le_logic_constraints(loc, k); }
