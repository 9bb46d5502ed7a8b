/* oracle/arithmetic_air.c -- AIR 4: ADD / SUB / LT / GT on 256-bit words (sixteen 16-bit limbs, one carry chain), 309
 * columns.  TEST INFRASTRUCTURE ONLY; "parity unpinned" by the reference (see gl.h): the reference proves its
 * arithmetic table through the out-of-tree plonky2_evm (call site plonky_block_proof_gen/src/proof_gen.rs:44-52,
 * table list prover_state.rs:85-93 "arithmetic", size range constants.rs:9); nothing under /root/reference shows its
 * columns.  Written from schoolbook addition with carries; the tests check the trace against Python's integers.
 *
 * Column map (shared with the product by specification):
 *   0..3 is_add, is_sub, is_lt, is_gt | 4..19 x limbs | 20..35 y limbs | 36..291 z bits (36 + 16 limb + bit) |
 *   292..307 carry out of each limb | 308 result (the comparison bit of lt / gt) */
#include "oracle.h"
#include <string.h>

enum { AR_OP = 0, AR_X = 4, AR_Y = 20, AR_Z = 36, AR_CARRY = 292, AR_RES = 308 };

static inline uint64_t smix(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* z = a + b or a - b over four 64-bit words; returns the carry / borrow out; also every 16-bit limb's carry out */
static unsigned add256(const uint64_t a[4], const uint64_t b[4], int subtract, uint64_t z[4], unsigned carries[16]) {
  unsigned c = 0;
  for (int k = 0; k < 16; k++) {
    const unsigned ak = (unsigned)(a[k / 4] >> (16 * (k % 4))) & 0xFFFF, bk = (unsigned)(b[k / 4] >> (16 * (k % 4))) & 0xFFFF;
    unsigned zk;
    if (!subtract) { unsigned s = ak + bk + c; zk = s & 0xFFFF; c = s >> 16; }
    else { int d = (int)ak - (int)bk - (int)c; c = d < 0; zk = (unsigned)(d + (c ? 65536 : 0)); }
    carries[k] = c;
    if (k % 4 == 0) z[k / 4] = 0;
    z[k / 4] |= (uint64_t)zk << (16 * (k % 4));
  }
  return c;
}

/* Witness: n = 2^log_n rows x 309 columns, column-major.  inputs: [n][9] = operation code (0 none, 1 add, 2 sub, 3 lt,
 * 4 gt; anything else: none), x (four u64, least significant first), y; or NULL: row r draws
 * code = smix(seed ^ (0xFE << 32) ^ r) % 5 and word w of operand j = smix(seed ^ ((1 + 4j + w) << 32) ^ r). */
void orc_arithmetic_trace(uint64_t seed, const uint64_t* inputs, unsigned log_n, gl_t* t) {
  const size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < n; r++) {
    uint64_t x[4], y[4], z[4] = {0, 0, 0, 0};
    unsigned carries[16] = {0}, out = 0;
    uint64_t code = inputs ? inputs[r * 9] : smix(seed ^ (0xFEULL << 32) ^ r) % 5;
    const unsigned op = code <= 4 ? (unsigned)code : 0;
    for (int w = 0; w < 4; w++) {
      x[w] = inputs ? inputs[r * 9 + 1 + w] : smix(seed ^ ((uint64_t)(1 + w) << 32) ^ r);
      y[w] = inputs ? inputs[r * 9 + 5 + w] : smix(seed ^ ((uint64_t)(5 + w) << 32) ^ r);
    }
    if (op == 1) add256(x, y, 0, z, carries);                 /* z = x + y */
    else if (op == 2) add256(x, y, 1, z, carries);            /* z = x - y */
    else if (op == 3) out = add256(x, y, 1, z, carries);      /* x < y: x - y borrows */
    else if (op == 4) out = add256(y, x, 1, z, carries);      /* x > y: y - x borrows */
#define PUT(col, v) t[(size_t)(col) * n + r] = (gl_t)(v)
    for (unsigned i = 0; i < 4; i++) PUT(AR_OP + i, op == i + 1);
    for (int k = 0; k < 16; k++) {
      PUT(AR_X + k, (x[k / 4] >> (16 * (k % 4))) & 0xFFFF);
      PUT(AR_Y + k, (y[k / 4] >> (16 * (k % 4))) & 0xFFFF);
      PUT(AR_CARRY + k, carries[k]);
    }
    for (int i = 0; i < 256; i++) PUT(AR_Z + i, (z[i / 64] >> (i % 64)) & 1);
    PUT(AR_RES, out);
#undef PUT
  }
}

/* ---- constraints, base field (the quotient on the LDE coset) ---- */
#define FT gl_t
#define FK(c) ((gl_t)(c))
#define FADD gl_add
#define FSUB gl_sub
#define FMUL gl_mul
#define FNAME(n) ab_##n
#define CONS_T orc_consumer
#define CONS_ALL(k, c) orc_cons(k, c)
#include "arithmetic_air_body.inc"
#undef FT
#undef FK
#undef FADD
#undef FSUB
#undef FMUL
#undef FNAME
#undef CONS_T
#undef CONS_ALL
void orc_arithmetic_constraints_base(const gl_t* loc, const gl_t* nxt, orc_consumer* k) { (void)nxt; ab_arithmetic_constraints(loc, k); }

/* ---- the same over the extension (the verifier's check at zeta) ---- */
#define FT gl2_t
#define FK(c) gl2_from((gl_t)(c))
#define FADD gl2_add
#define FSUB gl2_sub
#define FMUL gl2_mul
#define FNAME(n) ae_##n
#define CONS_T orc_consumer2
#define CONS_ALL(k, c) orc_cons2(k, c)
#include "arithmetic_air_body.inc"
void orc_arithmetic_constraints_ext(const gl2_t* loc, const gl2_t* nxt, orc_consumer2* k) { (void)nxt; ae_arithmetic_constraints(loc, k); }
