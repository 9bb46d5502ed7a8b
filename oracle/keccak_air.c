/* oracle/keccak_air.c -- AIR 1: one round of Keccak-f[1600] per trace row (24 rows per permutation), 2430 columns + the lookup's filter column (2430: g, ctl.c).
 * TEST INFRASTRUCTURE ONLY; "parity unpinned" by the reference (see gl.h): the reference proves its Keccak table
 * through the out-of-tree plonky2_evm (call site plonky_block_proof_gen/src/proof_gen.rs:44-52, table list
 * prover_state.rs:85-93, size range constants.rs:12); nothing under /root/reference shows its columns.  This AIR is
 * written from the public specification (FIPS 202 section 3.2) and pinned by Keccak known answers
 * (tests/test_keccak_air.py: the trace of a sponge block reproduces hashlib's SHA3-256).
 *
 * Column map (shared with the product by specification):
 *   0..23 round flags | 24..73 A limbs (24 + 2(x+5y) + h) | 74..393 C[x][z] | 394..713 C'[x][z] |
 *   714..2313 A'[x+5y][z] | 2314..2363 A'' limbs | 2364..2427 A''[0][0] bits | 2428..2429 A'''[0][0] limbs */
#include "oracle.h"
#include <string.h>

enum { KC_STEP = 0, KC_A = 24, KC_C = 74, KC_CP = 394, KC_AP = 714, KC_APP = 2314, KC_APP0 = 2364, KC_APPP = 2428 };

static const uint64_t KRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
/* rho offsets, KRHO[x][y] (FIPS 202 table 2) */
static const unsigned KRHO[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56},
                                    {27, 20, 39, 8, 14}};

static inline uint64_t rotl64(uint64_t v, unsigned r) { return r ? (v << r) | (v >> (64 - r)) : v; }
static inline uint64_t smix(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* One round on A[x][y]; writes the intermediate values of the row and advances A. */
typedef struct { uint64_t C[5], Cp[5], Ap[5][5], App[5][5], Appp00; } kround_t;
static void keccak_round(uint64_t A[5][5], unsigned rnd, kround_t* o) {
  uint64_t D[5], B[5][5];
  for (int x = 0; x < 5; x++) o->C[x] = A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4];
  for (int x = 0; x < 5; x++) {
    D[x] = o->C[(x + 4) % 5] ^ rotl64(o->C[(x + 1) % 5], 1);
    o->Cp[x] = o->C[x] ^ D[x];
    for (int y = 0; y < 5; y++) o->Ap[x][y] = A[x][y] ^ D[x];
  }
  for (int x = 0; x < 5; x++)
    for (int y = 0; y < 5; y++) B[y][(2 * x + 3 * y) % 5] = rotl64(o->Ap[x][y], KRHO[x][y]);
  for (int x = 0; x < 5; x++)
    for (int y = 0; y < 5; y++) o->App[x][y] = B[x][y] ^ (~B[(x + 1) % 5][y] & B[(x + 2) % 5][y]);
  o->Appp00 = o->App[0][0] ^ KRC[rnd];
  for (int x = 0; x < 5; x++)
    for (int y = 0; y < 5; y++) A[x][y] = o->App[x][y];
  A[0][0] = o->Appp00;
}
void orc_keccak_f(uint64_t lanes[25]) { /* lanes[x + 5y]; the permutation alone, for the known-answer tests */
  uint64_t A[5][5];
  kround_t o;
  for (int x = 0; x < 5; x++)
    for (int y = 0; y < 5; y++) A[x][y] = lanes[x + 5 * y];
  for (unsigned r = 0; r < 24; r++) keccak_round(A, r, &o);
  for (int x = 0; x < 5; x++)
    for (int y = 0; y < 5; y++) lanes[x + 5 * y] = A[x][y];
}

/* Witness: n = 2^log_n rows x 2431 columns (the last one, g, zero), column-major.  inputs: [ceil(n/24)][25] lanes (x + 5y) or NULL, then
 * lane l of permutation p is splitmix64(seed ^ (l << 32) ^ p). */
void orc_keccak_trace(uint64_t seed, const uint64_t* inputs, unsigned log_n, gl_t* t) {
  const size_t n = (size_t)1 << log_n, n_perm = (n + 23) / 24;
#pragma omp parallel for schedule(static)
  for (size_t p = 0; p < n_perm; p++) {
    uint64_t A[5][5];
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++)
        A[x][y] = inputs ? inputs[p * 25 + x + 5 * y] : smix(seed ^ ((uint64_t)(x + 5 * y) << 32) ^ p);
    for (unsigned rnd = 0; rnd < 24; rnd++) {
      const size_t r = p * 24 + rnd;
      if (r >= n) break;
      uint64_t in[5][5];
      memcpy(in, A, sizeof(in));
      kround_t o;
      keccak_round(A, rnd, &o);
#define PUT(col, v) t[(size_t)(col) * n + r] = (gl_t)(v)
      for (int i = 0; i < 24; i++) PUT(KC_STEP + i, i == (int)rnd);
      for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++) {
          const int l = x + 5 * y;
          PUT(KC_A + 2 * l, in[x][y] & 0xFFFFFFFFULL);
          PUT(KC_A + 2 * l + 1, in[x][y] >> 32);
          PUT(KC_APP + 2 * l, o.App[x][y] & 0xFFFFFFFFULL);
          PUT(KC_APP + 2 * l + 1, o.App[x][y] >> 32);
          for (int z = 0; z < 64; z++) PUT(KC_AP + 64 * l + z, (o.Ap[x][y] >> z) & 1);
        }
      for (int x = 0; x < 5; x++)
        for (int z = 0; z < 64; z++) {
          PUT(KC_C + 64 * x + z, (o.C[x] >> z) & 1);
          PUT(KC_CP + 64 * x + z, (o.Cp[x] >> z) & 1);
        }
      for (int z = 0; z < 64; z++) PUT(KC_APP0 + z, (o.App[0][0] >> z) & 1);
      PUT(KC_APPP, o.Appp00 & 0xFFFFFFFFULL);
      PUT(KC_APPP + 1, o.Appp00 >> 32);
      PUT(2430, 0); /* g: nothing exposed to the lookup (orc_ctl_set_filter marks the exposed rows, ctl.c) */
#undef PUT
    }
  }
}

/* ---- constraints, base field (the quotient on the LDE coset) ---- */
#define FT gl_t
#define FK(c) ((gl_t)(c))
#define FADD gl_add
#define FSUB gl_sub
#define FMUL gl_mul
#define FNAME(n) kb_##n
#define CONS_T orc_consumer
#define CONS_ALL(k, c) orc_cons(k, c)
#define CONS_TRANS(k, c) orc_cons(k, gl_mul(c, (k)->z_last))
#define CONS_FIRST(k, c) orc_cons(k, gl_mul(c, (k)->l_first))
#include "keccak_air_body.inc"
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
void orc_keccak_constraints_base(const gl_t* loc, const gl_t* nxt, orc_consumer* k) { kb_keccak_constraints(loc, nxt, k); }

/* ---- the same over the extension (the verifier's check at zeta) ---- */
#define FT gl2_t
#define FK(c) gl2_from((gl_t)(c))
#define FADD gl2_add
#define FSUB gl2_sub
#define FMUL gl2_mul
#define FNAME(n) ke_##n
#define CONS_T orc_consumer2
#define CONS_ALL(k, c) orc_cons2(k, c)
#define CONS_TRANS(k, c) orc_cons2(k, gl2_mul(c, (k)->z_last))
#define CONS_FIRST(k, c) orc_cons2(k, gl2_mul(c, (k)->l_first))
#include "keccak_air_body.inc"
void orc_keccak_constraints_ext(const gl2_t* loc, const gl2_t* nxt, orc_consumer2* k) { ke_keccak_constraints(loc, nxt, k); }
