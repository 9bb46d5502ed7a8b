/* oracle/memory_air.c -- AIR 3: a memory log sorted by (address, timestamp), one operation per trace row, 44 columns.
 * TEST INFRASTRUCTURE ONLY; "parity unpinned" by the reference (see gl.h): the reference proves its memory table
 * through the out-of-tree plonky2_evm (call site plonky_block_proof_gen/src/proof_gen.rs:44-52, table list
 * prover_state.rs:85-93 "memory", size range constants.rs:15); nothing under /root/reference shows its columns.
 * Written from the definition of a consistent memory: sorted by address then time, a read returns the value the
 * previous operation on the address left, memory starts as zeros.  The tests check the trace against a Python model.
 *
 * Column map (shared with the product by specification):
 *   0 is_read | 1 address | 2 timestamp | 3..10 value limbs | 11 address_changed (the NEXT row is on another address) |
 *   12..43 bits of the gap to the next row (address' - address - 1 across a change, timestamp' - timestamp - 1 within) */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

enum { MM_READ = 0, MM_ADDR = 1, MM_TS = 2, MM_VAL = 3, MM_CHG = 11, MM_GAP = 12 };

static inline uint64_t smix(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* Witness: n = 2^log_n rows x 44 columns, column-major.  inputs: [n][11] = is_read, address, timestamp, value[8],
 * sorted by the caller; or NULL: a log drawn from the seed, walked here in order (the product's kernel computes each
 * row independently): four operations per address, address of group g = 4g + (h(0xA0, g) & 3), timestamp of row
 * i = 8i + (h(0xA1, i) & 7), is_read = h(0xA2, i) & 1, a write stores limbs h(0xB0 + k, i) & 0xFFFFFFFF;
 * h(c, i) = smix(seed ^ (c << 32) ^ i). */
void orc_memory_trace(uint64_t seed, const uint64_t* inputs, unsigned log_n, gl_t* t) {
  const size_t n = (size_t)1 << log_n;
  uint64_t* log = (uint64_t*)malloc(n * 11 * sizeof(uint64_t));
  if (inputs) memcpy(log, inputs, n * 11 * sizeof(uint64_t));
  else {
    uint64_t cell[8] = {0};
    for (size_t i = 0; i < n; i++) {
      uint64_t* r = log + i * 11;
      if ((i & 3) == 0) memset(cell, 0, sizeof(cell)); /* a new address: zeros */
      r[0] = smix(seed ^ (0xA2ULL << 32) ^ i) & 1;
      r[1] = 4 * (i >> 2) + (smix(seed ^ (0xA0ULL << 32) ^ (i >> 2)) & 3);
      r[2] = 8 * i + (smix(seed ^ (0xA1ULL << 32) ^ i) & 7);
      if (!r[0])
        for (int k = 0; k < 8; k++) cell[k] = smix(seed ^ ((0xB0ULL + (uint64_t)k) << 32) ^ i) & 0xFFFFFFFFULL;
      memcpy(r + 3, cell, sizeof(cell));
    }
  }
  for (size_t i = 0; i < n; i++) {
    const uint64_t* r = log + i * 11;
    const int last = i + 1 == n, chg = !last && r[11 + 1] != r[1];
    const uint32_t gap = last ? 0 : (uint32_t)(chg ? r[11 + 1] - r[1] - 1 : r[11 + 2] - r[2] - 1);
#define PUT(col, v) t[(size_t)(col) * n + i] = gl_canon((gl_t)(v))
    PUT(MM_READ, r[0] & 1);
    PUT(MM_ADDR, r[1]);
    PUT(MM_TS, r[2]);
    for (int k = 0; k < 8; k++) PUT(MM_VAL + k, r[3 + k]);
    PUT(MM_CHG, chg);
    for (int z = 0; z < 32; z++) PUT(MM_GAP + z, (gap >> z) & 1);
    PUT(44, 0); /* g: nothing exposed to the lookup (orc_ctl_set_filter marks the exposed rows, ctl.c) */
#undef PUT
  }
  free(log);
}

/* ---- constraints, base field (the quotient on the LDE coset) ---- */
#define FT gl_t
#define FK(c) ((gl_t)(c))
#define FADD gl_add
#define FSUB gl_sub
#define FMUL gl_mul
#define FNAME(n) mb_##n
#define CONS_T orc_consumer
#define CONS_ALL(k, c) orc_cons(k, c)
#define CONS_TRANS(k, c) orc_cons(k, gl_mul(c, (k)->z_last))
#define CONS_FIRST(k, c) orc_cons(k, gl_mul(c, (k)->l_first))
#include "memory_air_body.inc"
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
void orc_memory_constraints_base(const gl_t* loc, const gl_t* nxt, orc_consumer* k) { mb_memory_constraints(loc, nxt, k); }

/* ---- the same over the extension (the verifier's check at zeta) ---- */
#define FT gl2_t
#define FK(c) gl2_from((gl_t)(c))
#define FADD gl2_add
#define FSUB gl2_sub
#define FMUL gl2_mul
#define FNAME(n) me_##n
#define CONS_T orc_consumer2
#define CONS_ALL(k, c) orc_cons2(k, c)
#define CONS_TRANS(k, c) orc_cons2(k, gl2_mul(c, (k)->z_last))
#define CONS_FIRST(k, c) orc_cons2(k, gl2_mul(c, (k)->l_first))
#include "memory_air_body.inc"
void orc_memory_constraints_ext(const gl2_t* loc, const gl2_t* nxt, orc_consumer2* k) { me_memory_constraints(loc, nxt, k); }
