/* oracle/gl.h -- Goldilocks field (p = 2^64 - 2^32 + 1) and its quadratic extension, plain C.
 *
 * TEST INFRASTRUCTURE ONLY.  This directory is the CPU restatement ("oracle") of the algorithms
 * reached through plonky_block_proof_gen/src/proof_gen.rs:44-52 (prove_root), :66-75, :97-103.
 * Those algorithms live in plonky2 @ 265d46a96ecfec49a32973f66f8aa811586c5d4a, which is NOT in
 * /root/reference, and the reference has no tests on this path (F5):
 *   ** parity unpinned ** by the reference.  What pins it instead is listed in DESIGN.md section 3.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything here.
 *
 * Field choice follows plonky_block_proof_gen/src/types.rs:10-18 (GoldilocksField,
 * PoseidonGoldilocksConfig, D = 2).  All stored values are canonical (< p).
 */
#ifndef ORACLE_GL_H
#define ORACLE_GL_H
#include <stdint.h>
#include <stddef.h>

typedef uint64_t gl_t;
typedef unsigned __int128 u128;
typedef struct { gl_t c0, c1; } gl2_t; /* c0 + c1*X, X^2 = 7 */

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL        /* 2^64 mod p */
#define GL_GENERATOR 7ULL           /* multiplicative generator; also the LDE coset shift */
#define GL_TWO_ADIC_ROOT 1753635133440165772ULL /* 7^((p-1)/2^32), order 2^32 */
#define GL_W 7ULL                   /* extension non-residue */

static inline gl_t gl_canon(gl_t a) { return a >= GL_P ? a - GL_P : a; }
static inline gl_t gl_add(gl_t a, gl_t b) { /* a,b canonical */
  gl_t s = a + b;
  if (s < a || s >= GL_P) s -= GL_P;
  return s;
}
static inline gl_t gl_sub(gl_t a, gl_t b) { return a >= b ? a - b : a + (GL_P - b); }
static inline gl_t gl_neg(gl_t a) { return a ? GL_P - a : 0; }
static inline gl_t gl_reduce128(u128 x) {
  /* x = lo + 2^64*(hh*2^32 + hl);  2^64 = 2^32-1, 2^96 = -1 (mod p) */
  uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
  uint64_t hh = hi >> 32, hl = hi & GL_EPS;
  uint64_t t0 = lo - hh;
  if (lo < hh) t0 -= GL_EPS;           /* borrow: subtract 2^64 = 2^32-1 more */
  uint64_t t1 = hl * GL_EPS;
  uint64_t r = t0 + t1;
  if (r < t1) r += GL_EPS;             /* carry */
  return gl_canon(r);
}
static inline gl_t gl_mul(gl_t a, gl_t b) { return gl_reduce128((u128)a * b); }
static inline gl_t gl_sqr(gl_t a) { return gl_mul(a, a); }
static inline gl_t gl_pow(gl_t a, uint64_t e) {
  gl_t r = 1;
  while (e) { if (e & 1) r = gl_mul(r, a); a = gl_sqr(a); e >>= 1; }
  return r;
}
static inline gl_t gl_inv(gl_t a) { return gl_pow(a, GL_P - 2); }
/* primitive 2^k-th root of unity, plonky2 convention: TWO_ADIC_ROOT^(2^(32-k)) */
static inline gl_t gl_root(unsigned k) {
  gl_t r = GL_TWO_ADIC_ROOT;
  for (unsigned i = k; i < 32; i++) r = gl_sqr(r);
  return r;
}

/* ---- quadratic extension ---- */
static inline gl2_t gl2(gl_t c0, gl_t c1) { gl2_t r = {c0, c1}; return r; }
static inline gl2_t gl2_from(gl_t a) { gl2_t r = {a, 0}; return r; }
static inline int gl2_eq(gl2_t a, gl2_t b) { return a.c0 == b.c0 && a.c1 == b.c1; }
static inline gl2_t gl2_add(gl2_t a, gl2_t b) { return gl2(gl_add(a.c0, b.c0), gl_add(a.c1, b.c1)); }
static inline gl2_t gl2_sub(gl2_t a, gl2_t b) { return gl2(gl_sub(a.c0, b.c0), gl_sub(a.c1, b.c1)); }
static inline gl2_t gl2_neg(gl2_t a) { return gl2(gl_neg(a.c0), gl_neg(a.c1)); }
static inline gl2_t gl2_mul(gl2_t a, gl2_t b) {
  gl_t c0 = gl_add(gl_mul(a.c0, b.c0), gl_mul(GL_W, gl_mul(a.c1, b.c1)));
  gl_t c1 = gl_add(gl_mul(a.c0, b.c1), gl_mul(a.c1, b.c0));
  return gl2(c0, c1);
}
static inline gl2_t gl2_scale(gl2_t a, gl_t s) { return gl2(gl_mul(a.c0, s), gl_mul(a.c1, s)); }
static inline gl2_t gl2_sqr(gl2_t a) { return gl2_mul(a, a); }
static inline gl2_t gl2_pow(gl2_t a, uint64_t e) {
  gl2_t r = gl2_from(1);
  while (e) { if (e & 1) r = gl2_mul(r, a); a = gl2_sqr(a); e >>= 1; }
  return r;
}
static inline gl2_t gl2_inv(gl2_t a) { /* 1/(c0 + c1 X) = (c0 - c1 X)/(c0^2 - 7 c1^2) */
  gl_t n = gl_sub(gl_sqr(a.c0), gl_mul(GL_W, gl_sqr(a.c1)));
  gl_t ni = gl_inv(n);
  return gl2(gl_mul(a.c0, ni), gl_mul(gl_neg(a.c1), ni));
}

static inline uint32_t bitrev32(uint32_t x, unsigned bits) {
  if (!bits) return 0;
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
  x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
  x = (x >> 16) | (x << 16);
  return x >> (32 - bits);
}
#endif
