/* oracle/byte_packing_air.c -- AIR 5: a big-endian sequence of 1..32 bytes and the 256-bit word it spells, one per
 * trace row, 299 columns.  TEST INFRASTRUCTURE ONLY; "parity unpinned" by the reference (see gl.h): the reference
 * proves its byte-packing table through the out-of-tree plonky2_evm (call site
 * plonky_block_proof_gen/src/proof_gen.rs:44-52, table list prover_state.rs:85-93 "byte_packing", size range
 * constants.rs:10); nothing under /root/reference shows its columns.  Written from what MLOAD_32BYTES / MSTORE_32BYTES
 * mean; the tests check the trace against int.from_bytes(..., "big").
 *
 * Column map (shared with the product by specification, AIRS.md section 2):
 *   0 is_read | 1..32 length flags (column j: len = j) | 33..288 bits of the 32 byte slots (33 + 8 slot + bit) |
 *   289..296 value limbs (u32, least significant first) | 297 address, 298 timestamp of the memory operation that moves
 *   the word (free columns: the lookup byte_packing -> memory of ctl.c reads them) */
#include "oracle.h"
#include <string.h>

enum { BP_READ = 0, BP_LEN = 1, BP_BITS = 33, BP_VAL = 289, BP_ADDR = 297, BP_TS = 298 };

static inline uint64_t smix(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* Witness: n = 2^log_n rows x 299 columns, column-major.  inputs: [n][6] = word 0: is_read (bit 0) | timestamp << 8,
 * word 1: len (low byte; 0 = padding row; above 32: 32) | address << 8 (32 bits each), the 32 byte slots as four u64 (slot i = byte i % 8 of word i / 8; slots from len on are ignored); or NULL: row
 * r draws is_read = h(0xC0) & 1, len = h(0xC1) % 33, word w = h(0xC2 + w), h(c) = smix(seed ^ (c << 32) ^ r), and sits
 * at address r with timestamp r + 2. */
void orc_byte_packing_trace(uint64_t seed, const uint64_t* inputs, unsigned log_n, gl_t* t) {
  const size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < n; r++) {
    const uint64_t rd = (inputs ? inputs[r * 6] : smix(seed ^ (0xC0ULL << 32) ^ r)) & 1;
    uint64_t len = inputs ? (inputs[r * 6 + 1] & 0xFF) : smix(seed ^ (0xC1ULL << 32) ^ r) % 33;
    if (len > 32) len = 32;
    const uint64_t address = inputs ? (inputs[r * 6 + 1] >> 8) & 0xFFFFFFFFULL : (uint64_t)r;
    const uint64_t timestamp = inputs ? (inputs[r * 6] >> 8) & 0xFFFFFFFFULL : (uint64_t)r + 2;
    unsigned char bytes[32];
    for (int s = 0; s < 32; s++) {
      const uint64_t w = inputs ? inputs[r * 6 + 2 + s / 8] : smix(seed ^ ((0xC2ULL + (uint64_t)(s / 8)) << 32) ^ r);
      bytes[s] = (uint64_t)s < len ? (unsigned char)(w >> (8 * (s % 8))) : 0;
    }
    /* the value, big-endian: bytes[0] is the most significant of the len bytes */
    unsigned char le[32] = {0}; /* little-endian image of the 256-bit word */
    for (uint64_t s = 0; s < len; s++) le[len - 1 - s] = bytes[s];
#define PUT(col, v) t[(size_t)(col) * n + r] = (gl_t)(v)
    PUT(BP_READ, rd);
    for (unsigned j = 1; j <= 32; j++) PUT(BP_LEN + j - 1, len == j);
    for (int s = 0; s < 32; s++)
      for (int b = 0; b < 8; b++) PUT(BP_BITS + 8 * s + b, (bytes[s] >> b) & 1);
    for (int k = 0; k < 8; k++)
      PUT(BP_VAL + k, (uint32_t)le[4 * k] | ((uint32_t)le[4 * k + 1] << 8) | ((uint32_t)le[4 * k + 2] << 16) | ((uint32_t)le[4 * k + 3] << 24));
    PUT(BP_ADDR, address);
    PUT(BP_TS, timestamp);
#undef PUT
  }
}

/* ---- constraints, base field (the quotient on the LDE coset) ---- */
#define FT gl_t
#define FK(c) ((gl_t)(c))
#define FADD gl_add
#define FSUB gl_sub
#define FMUL gl_mul
#define FNAME(n) pb_##n
#define CONS_T orc_consumer
#define CONS_ALL(k, c) orc_cons(k, c)
#include "byte_packing_air_body.inc"
#undef FT
#undef FK
#undef FADD
#undef FSUB
#undef FMUL
#undef FNAME
#undef CONS_T
#undef CONS_ALL
void orc_byte_packing_constraints_base(const gl_t* loc, const gl_t* nxt, orc_consumer* k) { (void)nxt; pb_byte_packing_constraints(loc, k); }

/* ---- the same over the extension (the verifier's check at zeta) ---- */
#define FT gl2_t
#define FK(c) gl2_from((gl_t)(c))
#define FADD gl2_add
#define FSUB gl2_sub
#define FMUL gl2_mul
#define FNAME(n) pe_##n
#define CONS_T orc_consumer2
#define CONS_ALL(k, c) orc_cons2(k, c)
#include "byte_packing_air_body.inc"
void orc_byte_packing_constraints_ext(const gl2_t* loc, const gl2_t* nxt, orc_consumer2* k) { (void)nxt; pe_byte_packing_constraints(loc, k); }
