/* oracle/keccak_sponge_air.c -- AIR 6: the absorbing side of Keccak-256, one 136-byte block per trace row, 2414
 * columns.  TEST INFRASTRUCTURE ONLY; "parity unpinned" by the reference (see gl.h): the reference proves its Keccak
 * sponge table through the out-of-tree plonky2_evm (call site plonky_block_proof_gen/src/proof_gen.rs:44-52, table
 * list prover_state.rs:85-93 "keccak_sponge", size range constants.rs:13); nothing under /root/reference shows its
 * columns.  Written from FIPS 202 (sponge construction, pad10*1, rate 1088); the tests check the rows of a message
 * against hashlib.
 *
 * Column map (shared with the product by specification):
 *   0 is_full | 1 is_final | 2..137 final-length flags | 138..1225 bits of the block as absorbed (138 + 8 byte + bit) |
 *   1226..2313 bits of the rate before the block (1226 + 32 limb + bit) | 2314..2329 capacity limbs before |
 *   2330..2363 rate limbs after the XOR | 2364..2413 state limbs after the permutation (not constrained here) */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

enum { SP_FULL = 0, SP_FINAL = 1, SP_LEN = 2, SP_BLOCK = 138, SP_RATE = 1226, SP_CAP = 2314, SP_XORED = 2330, SP_UPDATED = 2364 };

static inline uint64_t smix(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* The rows of one message (what the product's bp_keccak256_sponge_rows returns): per block 44 words = flags (1 full,
 * 2 final), message bytes in the block, the block as absorbed (17 words), the state before it (25 lanes).  Returns
 * the number of rows; rows may be NULL to count.  digest (32 bytes) may be NULL. */
size_t orc_keccak_sponge_rows(const uint8_t* msg, size_t len, uint64_t* rows, uint8_t* digest) {
  uint64_t st[25];
  memset(st, 0, sizeof(st));
  size_t n = 0;
  for (;;) {
    const int last = len < 136;
    uint8_t block[136];
    memset(block, 0, 136);
    const size_t take = last ? len : 136;
    if (take) memcpy(block, msg, take);
    if (last) { block[len] ^= 0x01; block[135] ^= 0x80; }
    if (rows) {
      uint64_t* r = rows + n * 44;
      r[0] = last ? 2 : 1;
      r[1] = take;
      memcpy(r + 2, block, 136);
      memcpy(r + 19, st, 200);
    }
    n++;
    uint64_t w[17];
    memcpy(w, block, 136);
    for (int i = 0; i < 17; i++) st[i] ^= w[i];
    orc_keccak_f(st);
    if (last) break;
    msg += 136;
    len -= 136;
  }
  if (digest) memcpy(digest, st, 32);
  return n;
}

/* Witness: n = 2^log_n rows x 2414 columns, column-major.  inputs: [n][44] as above (flags 0: a padding row); or NULL:
 * row r is a single-block message drawn from the seed -- h(c) = smix(seed ^ (c << 32) ^ r); h(0xD3) % 8 == 0: padding
 * row; else len = h(0xD0) % 136, message word w = h(0xD1 + (w << 8)), state before = 0. */
void orc_keccak_sponge_trace(uint64_t seed, const uint64_t* inputs, unsigned log_n, gl_t* t) {
  orc_keccak_sponge_trace_limit(seed, inputs, log_n, (size_t)-1, t);
}
/* row_limit: a seeded table asks for no more permutations than the transaction's Keccak-f table holds in full (ctl.c):
 * seeded rows from row_limit on are padding rows */
void orc_keccak_sponge_trace_limit(uint64_t seed, const uint64_t* inputs, unsigned log_n, size_t row_limit, gl_t* t) {
  const size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < n; r++) {
    uint64_t flags, len, blk[17], st[25], in[25], out[25];
    if (inputs) {
      const uint64_t* q = inputs + r * 44;
      flags = q[0] & 3; len = q[1];
      memcpy(blk, q + 2, 136);
      memcpy(st, q + 19, 200);
    } else {
      memset(st, 0, sizeof(st));
      memset(blk, 0, sizeof(blk));
      if (smix(seed ^ (0xD3ULL << 32) ^ r) % 8 == 0 || r >= row_limit) { flags = 0; len = 0; }
      else {
        flags = 2; len = smix(seed ^ (0xD0ULL << 32) ^ r) % 136;
        uint8_t* b = (uint8_t*)blk;
        for (uint64_t w = 0; w < 17; w++) {
          const uint64_t m = smix(seed ^ ((0xD1ULL + (w << 8)) << 32) ^ r);
          for (int j = 0; j < 8; j++)
            if (8 * w + (uint64_t)j < len) b[8 * w + j] = (uint8_t)(m >> (8 * j));
        }
        b[len] ^= 0x01;
        b[135] ^= 0x80;
      }
    }
    if (flags == 3) flags = 0;
    for (int l = 0; l < 25; l++) in[l] = l < 17 ? st[l] ^ blk[l] : st[l];
    memcpy(out, in, sizeof(out));
    if (flags) orc_keccak_f(out);
    else memset(out, 0, sizeof(out));
#define PUT(col, v) t[(size_t)(col) * n + r] = (gl_t)(v)
    PUT(SP_FULL, flags == 1);
    PUT(SP_FINAL, flags == 2);
    for (unsigned j = 0; j < 136; j++) PUT(SP_LEN + j, flags == 2 && len == j);
    for (int z = 0; z < 1088; z++) {
      PUT(SP_BLOCK + z, (blk[z / 64] >> (z % 64)) & 1);
      PUT(SP_RATE + z, (st[z / 64] >> (z % 64)) & 1);
    }
    for (int k = 0; k < 16; k++) PUT(SP_CAP + k, (st[17 + k / 2] >> (32 * (k % 2))) & 0xFFFFFFFFULL);
    for (int k = 0; k < 34; k++) PUT(SP_XORED + k, (in[k / 2] >> (32 * (k % 2))) & 0xFFFFFFFFULL);
    for (int k = 0; k < 50; k++) PUT(SP_UPDATED + k, (out[k / 2] >> (32 * (k % 2))) & 0xFFFFFFFFULL);
#undef PUT
  }
}

/* ---- constraints, base field (the quotient on the LDE coset) ---- */
#define FT gl_t
#define FK(c) ((gl_t)(c))
#define FADD gl_add
#define FSUB gl_sub
#define FMUL gl_mul
#define FNAME(n) sb_##n
#define CONS_T orc_consumer
#define CONS_ALL(k, c) orc_cons(k, c)
#define CONS_TRANS(k, c) orc_cons(k, gl_mul(c, (k)->z_last))
#define CONS_FIRST(k, c) orc_cons(k, gl_mul(c, (k)->l_first))
#include "keccak_sponge_air_body.inc"
#undef FT
#undef FK
#undef FADD
#undef FSUB
#undef FMUL
#undef FNAME
#undef CONS_T
#undef CONS_ALL
#undef CONS_TRANS
#undef CONS_FIRST
void orc_keccak_sponge_constraints_base(const gl_t* loc, const gl_t* nxt, orc_consumer* k) { sb_sponge_constraints(loc, nxt, k); }

/* ---- the same over the extension (the verifier's check at zeta) ---- */
#define FT gl2_t
#define FK(c) gl2_from((gl_t)(c))
#define FADD gl2_add
#define FSUB gl2_sub
#define FMUL gl2_mul
#define FNAME(n) se_##n
#define CONS_T orc_consumer2
#define CONS_ALL(k, c) orc_cons2(k, c)
#define CONS_TRANS(k, c) orc_cons2(k, gl2_mul(c, (k)->z_last))
#define CONS_FIRST(k, c) orc_cons2(k, gl2_mul(c, (k)->l_first))
#include "keccak_sponge_air_body.inc"
void orc_keccak_sponge_constraints_ext(const gl2_t* loc, const gl2_t* nxt, orc_consumer2* k) { se_sponge_constraints(loc, nxt, k); }
