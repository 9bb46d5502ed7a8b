/* oracle/arithmetic_mul_air.c -- AIR 7: x * y = z + 2^256 w on 256-bit words, one product per trace row, 1217 columns.
 * TEST INFRASTRUCTURE ONLY; "parity unpinned" by the reference (see gl.h): the multiplicative half of the arithmetic
 * table the reference proves through the out-of-tree plonky2_evm (call site
 * plonky_block_proof_gen/src/proof_gen.rs:44-52, table list prover_state.rs:85-93 "arithmetic"); nothing under
 * /root/reference shows its columns.  Written from schoolbook multiplication; the product itself is computed here on
 * 64-bit words with unsigned __int128 (the product's kernel works limb by limb), the tests check it against Python's
 * integers.
 *
 * Column map (shared with the product by specification):
 *   0 is_mul | 1..16 x limbs | 17..32 y limbs | 33..288 z bits (33 + 16 limb + bit) | 289..544 w bits |
 *   545..1216 carry bits (545 + 21 column + bit): the carry out of product column 0..31 */
#include "oracle.h"
#include <string.h>

enum { AM_MUL = 0, AM_X = 1, AM_Y = 17, AM_Z = 33, AM_W = 289, AM_CARRY = 545 };

static inline uint64_t smix(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* Witness: n = 2^log_n rows x 1217 columns, column-major.  inputs: [n][9] = is_mul, x (four u64, least significant
 * first), y; or NULL: row r draws is_mul = smix(seed ^ (0xE0 << 32) ^ r) % 4 != 0 and word w of operand j =
 * smix(seed ^ ((1 + 4j + w) << 32) ^ r). */
void orc_arithmetic_mul_trace(uint64_t seed, const uint64_t* inputs, unsigned log_n, gl_t* t) {
  const size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < n; r++) {
    uint64_t x[4], y[4], prod[8] = {0};
    const int mul = inputs ? (int)(inputs[r * 9] & 1) : smix(seed ^ (0xE0ULL << 32) ^ r) % 4 != 0;
    for (int w = 0; w < 4; w++) {
      x[w] = inputs ? inputs[r * 9 + 1 + w] : smix(seed ^ ((uint64_t)(1 + w) << 32) ^ r);
      y[w] = inputs ? inputs[r * 9 + 5 + w] : smix(seed ^ ((uint64_t)(5 + w) << 32) ^ r);
    }
    if (mul) /* the 512-bit product on 64-bit words */
      for (int i = 0; i < 4; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < 4; j++) {
          c += (unsigned __int128)x[i] * y[j] + prod[i + j];
          prod[i + j] = (uint64_t)c;
          c >>= 64;
        }
        prod[i + 4] = (uint64_t)c;
      }
#define LIMB(a, k) (((a)[(k) / 4] >> (16 * ((k) % 4))) & 0xFFFF)
#define PUT(col, v) t[(size_t)(col) * n + r] = (gl_t)(v)
    PUT(AM_MUL, mul);
    for (int k = 0; k < 16; k++) { PUT(AM_X + k, LIMB(x, k)); PUT(AM_Y + k, LIMB(y, k)); }
    for (int i = 0; i < 256; i++) {
      PUT(AM_Z + i, (prod[i / 64] >> (i % 64)) & 1);
      PUT(AM_W + i, (prod[4 + i / 64] >> (i % 64)) & 1);
    }
    /* the carries are what the column equations leave no choice about */
    uint64_t carry = 0;
    for (int k = 0; k < 32; k++) {
      uint64_t sum = carry;
      if (mul)
        for (int i = 0; i < 16; i++)
          if (k - i >= 0 && k - i < 16) sum += LIMB(x, i) * LIMB(y, k - i);
      carry = (sum - LIMB(prod, k)) >> 16;
      for (int j = 0; j < 21; j++) PUT(AM_CARRY + 21 * k + j, (carry >> j) & 1);
    }
#undef PUT
#undef LIMB
  }
}

/* ---- constraints, base field (the quotient on the LDE coset) ---- */
#define FT gl_t
#define FK(c) ((gl_t)(c))
#define FADD gl_add
#define FSUB gl_sub
#define FMUL gl_mul
#define FNAME(n) mb2_##n
#define CONS_T orc_consumer
#define CONS_ALL(k, c) orc_cons(k, c)
#include "arithmetic_mul_air_body.inc"
#undef FT
#undef FK
#undef FADD
#undef FSUB
#undef FMUL
#undef FNAME
#undef CONS_T
#undef CONS_ALL
void orc_arithmetic_mul_constraints_base(const gl_t* loc, const gl_t* nxt, orc_consumer* k) { (void)nxt; mb2_mul_constraints(loc, k); }

/* ---- the same over the extension (the verifier's check at zeta) ---- */
#define FT gl2_t
#define FK(c) gl2_from((gl_t)(c))
#define FADD gl2_add
#define FSUB gl2_sub
#define FMUL gl2_mul
#define FNAME(n) me2_##n
#define CONS_T orc_consumer2
#define CONS_ALL(k, c) orc_cons2(k, c)
#include "arithmetic_mul_air_body.inc"
void orc_arithmetic_mul_constraints_ext(const gl2_t* loc, const gl2_t* nxt, orc_consumer2* k) { (void)nxt; me2_mul_constraints(loc, k); }
