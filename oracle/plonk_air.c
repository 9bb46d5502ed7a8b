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
 * Constants (85): 0 q_arith, 1 q_sbox, 2 c0, 3 c1, 4 q_hash, 5 + j sigma_j.  Copy permutation: wire (j, i) is the field element
 * 7^j w^i; sigma maps every wire to the next one of its equivalence class.  The circuit: rows in groups of four,
 *   4g   arith, free inputs     4g+1  arith on the outputs of 4g     4g+2  S-box of the first 11 outputs of 4g+1
 *   4g+3 arith whose first 11 `a` inputs are the S-box outputs;  group 0 = the public-input row and three no-ops;
 * c_j of row 12 (j < 4) is the public input w_j of row 0, which is word j of the in-circuit hash (see below).  Equivalence classes are listed below as explicit sets; the
 * product (csrc/air.hpp, plonk::sigma_of) computes "the next member" in closed form. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

enum { PW = 135, PK = 85, ROUTED = 80, SLOTS = 20, SBOX = 11, SB0 = 80, SIG0 = 5 /* first sigma column */ };
/* Round 5: the circuit hashes its public-input list in-circuit.  Rows 4 .. 4 + H - 1 (H = ceil(len / 8)) are Poseidon
 * rows (constant column 4 = q_hash): one permutation per row, wires 0..11 state in, 12..23 state out, 24..59 the S-box
 * inputs of rounds 1..3, 60..81 word 0 going into the S-box of rounds 4..25, 82..129 the S-box inputs of rounds 26..29.
 * Chunk h of the list sits in wires 0..7 of row 4 + h; the rest of the incoming state is copied from the previous row's
 * output, or (row 4) from the d wires of row 1, an arithmetic row whose gate constants are zero.  The first four output
 * words of the last hash row are the public-input wires of row 0.  Arithmetic groups start at row 12. */
/* Round 5, second step: the Poseidon gate has upstream's swap (wire 130 = a bit s, wires 131..134 = delta_i = s (in[4+i] -
 * in[i]); the permutation runs on (in[i] + delta_i, in[4+i] - delta_i, in[8..11])), and a circuit may walk Merkle paths with
 * it: rows 17 .. 17 + n_paths * depth - 1, one level per row -- node in 0..3, sibling in 4..7, zeros in 8..11, position bit
 * on the swap wire, the node above in out 0..3.  Path p's leaf digest is words path_pi0 + 8p .. + 3 of the public-input
 * list, the cap entry it arrives at words path_pi0 + 8p + 4 .. + 7.  Arithmetic groups start at the first multiple of four
 * past the Merkle rows. */
enum { HROW0 = 4, MROW0 = 17 /* after at most 13 rows of list */, ZROW = 1, HIN = 0, HOUT = 12, HF1 = 24, HPART = 60, HF2 = 82, HSWAP = 130, HDELTA = 131, HWIRES = 135 };
/* leaf_len > 0: after the Merkle rows every path has ceil(leaf_len / 8) sponge rows that hash the row its leaf digest is the
 * digest of (the child's opened trace row: free wires); their last output is the path's first node and the list's leaf words */
static unsigned arith_row0_of(unsigned n_paths, unsigned depth, unsigned leaf_len) {
  return (MROW0 + n_paths * depth + (leaf_len ? n_paths * ((leaf_len + 7) / 8) : 0) + 3) / 4 * 4;
}
static const gl_t PRC[360] = {
#include "poseidon_rc.inc"
};
static gl_t mds_at(int r, int c) { /* out[r] = sum_c M[r][c] in[c]: the circulant with first ROW (17, 15, 41, ...), plus 8 at [0][0] */
  static const gl_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  return CIRC[((c - r) % 12 + 12) % 12] + (r == 0 && c == 0 ? 8 : 0);
}
static unsigned hash_rows_of(unsigned pi_len) { return (pi_len + 7) / 8; }

static inline uint64_t smix(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static inline gl_t rnd(uint64_t seed, uint64_t col, uint64_t row) { return gl_canon(smix(seed ^ (col << 32) ^ row)); }

void orc_stark_public_input_list(uint64_t seed, gl_t out[4]) { /* of a lone table proof: four words from its seed */
  for (uint64_t j = 0; j < 4; j++) out[j] = gl_canon(smix(seed ^ ((0x50 + j) << 32)));
}
void orc_stark_public_inputs(uint64_t seed, gl_t out[4]) { /* ... and what its first row is bound to: the hash of that list */
  gl_t pi[4];
  orc_stark_public_input_list(seed, pi);
  orc_hash_no_pad(pi, 4, out);
}

typedef struct { uint32_t col, row; } wire_t;
/* ties the wires of one class into a cycle: each member's sigma is the next member, the last one's the first */
static void tie(gl_t* consts, size_t N, const gl_t* kpow, const gl_t* wpow, const wire_t* m, int n) {
  for (int i = 0; i < n; i++) {
    const wire_t to = m[(i + 1) % n];
    consts[(size_t)(SIG0 + m[i].col) * N + m[i].row] = gl_mul(kpow[to.col], wpow[to.row]);
  }
}
void orc_plonk_constants(uint64_t seed, unsigned log_n, unsigned pi_len, unsigned n_paths, unsigned depth, unsigned path_pi0,
                         unsigned leaf_len, gl_t* consts) {
  const size_t N = (size_t)1 << log_n;
  const unsigned H = hash_rows_of(pi_len), last_len = pi_len - 8 * (H - 1);
  const unsigned AROW0 = arith_row0_of(n_paths, depth, leaf_len), MROWS = n_paths * depth;
  const unsigned LH = leaf_len ? (leaf_len + 7) / 8 : 0, LROW0 = MROW0 + MROWS, LROWS = n_paths * LH;
  gl_t *kpow = (gl_t*)malloc(ROUTED * sizeof(gl_t)), *wpow = (gl_t*)malloc(N * sizeof(gl_t));
  kpow[0] = 1;
  for (int j = 1; j < ROUTED; j++) kpow[j] = gl_mul(kpow[j - 1], 7);
  wpow[0] = 1;
  for (size_t i = 1; i < N; i++) wpow[i] = gl_mul(wpow[i - 1], gl_root(log_n));
  for (size_t i = 0; i < N; i++) {
    const int gate_row = i >= AROW0;
    consts[0 * N + i] = (gate_row && (i % 4 != 2)) || i == ZROW; /* row 1: an arithmetic gate with zero constants: d = 0 */
    consts[1 * N + i] = gate_row && (i % 4 == 2);
    consts[2 * N + i] = i == ZROW ? 0 : rnd(seed ^ 0xC0115700C0115700ULL, 2, i);
    consts[3 * N + i] = i == ZROW ? 0 : rnd(seed ^ 0xC0115700C0115700ULL, 3, i);
    consts[4 * N + i] = (i >= HROW0 && i < HROW0 + H) || (i >= MROW0 && i < MROW0 + MROWS + LROWS);
    for (int j = 0; j < ROUTED; j++) consts[(size_t)(SIG0 + j) * N + i] = gl_mul(kpow[j], wpow[i]); /* untied: itself */
  }
  /* the sponge: which incoming state words of hash row h are NOT words of the list */
  {
    unsigned zero_wire = 0;
    for (unsigned h = 0; h < H; h++)
      for (unsigned k = 0; k < 12; k++) {
        const int carried = k >= 8 || (h == H - 1 && k >= last_len);
        if (!carried) continue;
        if (h == 0) { /* nothing before the first chunk: the word is zero -- a copy of one of row 1's d wires */
          const wire_t z[2] = {{HIN + k, HROW0}, {4 * zero_wire + 3, ZROW}};
          tie(consts, N, kpow, wpow, z, 2);
          zero_wire++;
        } else { /* what the previous permutation left there */
          const wire_t c[2] = {{HIN + k, HROW0 + h}, {HOUT + k, HROW0 + h - 1}};
          tie(consts, N, kpow, wpow, c, 2);
        }
      }
  }
  /* the Merkle paths: level by level the node climbs; the ends are words of the list; every capacity word is zero */
  if (MROWS) {
    wire_t* zeros = (wire_t*)malloc((4 * (MROWS + n_paths) + 1) * sizeof(wire_t));
    int nz = 0;
    zeros[nz++] = (wire_t){4 * 19 + 3, ZROW}; /* the last zero wire of row 1 */
    for (unsigned p = 0; p < n_paths; p++)
      for (unsigned l = 0; l < depth; l++) {
        const uint32_t row = MROW0 + p * depth + l;
        for (uint32_t j = 0; j < 4; j++) {
          zeros[nz++] = (wire_t){HIN + 8 + j, row};
          if (l == 0 && leaf_len) { /* ... which the path's own sponge rows compute: list word, first node, last sponge output */
            const unsigned i = path_pi0 + 8 * p + j;
            const wire_t leaf[3] = {{HIN + i % 8, HROW0 + i / 8}, {HIN + j, row}, {HOUT + j, LROW0 + p * LH + LH - 1}};
            tie(consts, N, kpow, wpow, leaf, 3);
          } else if (l == 0) { /* the leaf digest: word path_pi0 + 8p + j of the list, wherever the sponge absorbs it */
            const unsigned i = path_pi0 + 8 * p + j;
            const wire_t leaf[2] = {{HIN + j, row}, {HIN + i % 8, HROW0 + i / 8}};
            tie(consts, N, kpow, wpow, leaf, 2);
          } else {
            const wire_t up[2] = {{HIN + j, row}, {HOUT + j, row - 1}};
            tie(consts, N, kpow, wpow, up, 2);
          }
          if (l == depth - 1) { /* the cap entry */
            const unsigned i = path_pi0 + 8 * p + 4 + j;
            const wire_t top[2] = {{HOUT + j, row}, {HIN + i % 8, HROW0 + i / 8}};
            tie(consts, N, kpow, wpow, top, 2);
          }
        }
      }
    /* the leaf sponges: row h of path p absorbs eight words of the opened row; what it does not take from the row it
     * carries from the row before (h > 0) or, in its first row, is zero (the capacity words: more members of the zero class) */
    for (unsigned p = 0; p < n_paths && leaf_len; p++)
      for (unsigned h = 0; h < LH; h++) {
        const uint32_t row = LROW0 + p * LH + h;
        const unsigned last_len = leaf_len - 8 * (LH - 1);
        for (uint32_t k = 0; k < 12; k++) {
          const int carried = k >= 8 || (h == LH - 1 && k >= last_len);
          if (!carried) continue;
          if (h == 0) zeros[nz++] = (wire_t){HIN + k, row};
          else {
            const wire_t c[2] = {{HIN + k, row}, {HOUT + k, row - 1}};
            tie(consts, N, kpow, wpow, c, 2);
          }
        }
      }
    tie(consts, N, kpow, wpow, zeros, nz);
    free(zeros);
  }
  for (size_t g = AROW0 / 4; g < N / 4; g++) {
    const uint32_t r = (uint32_t)(4 * g);
    for (uint32_t s = 0; s < SLOTS; s++) {
      /* d_s(4g) is the `a` input of slot s and the `b` input of slot s - 1 in the next row */
      const wire_t dab[3] = {{4 * s + 3, r}, {4 * s, r + 1}, {4 * ((s + SLOTS - 1) % SLOTS) + 1, r + 1}};
      tie(consts, N, kpow, wpow, dab, 3);
      if (r == AROW0 && s < 4) { /* c_s of the first computing rows is public input s, which is word s of the hash */
        const wire_t cc[4] = {{s, 0}, {4 * s + 2, r}, {4 * s + 2, r + 1}, {HOUT + s, HROW0 + H - 1}};
        tie(consts, N, kpow, wpow, cc, 4);
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

/* Witness: row by row.  consts: the circuit's constants (c0, c1 are read).  pi: the public-input list the hash rows
 * absorb; the four public inputs of row 0 are its hash, computed here the way the circuit computes it. */
static gl_t pow7(gl_t x) { const gl_t x2 = gl_mul(x, x), x3 = gl_mul(x2, x), x6 = gl_mul(x3, x3); return gl_mul(x6, x); }
static void mds_layer(gl_t st[12]) {
  gl_t o[12];
  for (int r = 0; r < 12; r++) {
    gl_t acc = 0;
    for (int c = 0; c < 12; c++) acc = gl_add(acc, gl_mul(mds_at(r, c), st[c]));
    o[r] = acc;
  }
  memcpy(st, o, sizeof(o));
}
/* one permutation, every S-box input kept in the row's wires */
static void permute_into_row(gl_t st[12], gl_t* t, size_t N, size_t row) {
#define W(col, row) t[(size_t)(col) * N + (row)]
  for (int rnd_no = 0; rnd_no < 30; rnd_no++) {
    for (int k = 0; k < 12; k++) st[k] = gl_add(st[k], PRC[12 * rnd_no + k]);
    const int full = rnd_no < 4 || rnd_no >= 26;
    if (rnd_no >= 1 && rnd_no <= 3) for (int k = 0; k < 12; k++) W(HF1 + 12 * (rnd_no - 1) + k, row) = st[k];
    if (rnd_no >= 4 && rnd_no <= 25) W(HPART + rnd_no - 4, row) = st[0];
    if (rnd_no >= 26) for (int k = 0; k < 12; k++) W(HF2 + 12 * (rnd_no - 26) + k, row) = st[k];
    if (full) for (int k = 0; k < 12; k++) st[k] = pow7(st[k]);
    else st[0] = pow7(st[0]);
    mds_layer(st);
  }
  for (int k = 0; k < 12; k++) W(HOUT + k, row) = st[k];
#undef W
}
/* paths (n_paths > 0): per path 1 + 4 depth words: the leaf's position, then the siblings from the leaf upward */
void orc_plonk_trace(uint64_t seed, const gl_t* pi, unsigned pi_len, unsigned n_paths, unsigned depth, unsigned path_pi0,
                     unsigned leaf_len, const gl_t* paths, const gl_t* consts, unsigned log_n, gl_t* t) {
  const size_t N = (size_t)1 << log_n;
  const unsigned H = hash_rows_of(pi_len);
  const unsigned AROW0 = arith_row0_of(n_paths, depth, leaf_len);
  const unsigned LH = leaf_len ? (leaf_len + 7) / 8 : 0, LROW0 = MROW0 + n_paths * depth;
  const size_t path_words = 1 + 4 * (size_t)depth + leaf_len;
#define W(col, row) t[(size_t)(col) * N + (row)]
  for (size_t i = 0; i < N; i++) /* every wire starts free; the rows below overwrite what the circuit computes */
    for (int c = 0; c < PW; c++) W(c, i) = rnd(seed, c, i);
  for (int s = 0; s < SLOTS; s++) W(4 * s + 3, ZROW) = 0; /* 0 a b + 0 c */
  /* the hash rows: absorb eight words, permute, keep every S-box input on the way */
  gl_t st[12] = {0};
  for (unsigned h = 0; h < H; h++) {
    const size_t row = HROW0 + h;
    const unsigned take = pi_len - 8 * h < 8 ? pi_len - 8 * h : 8;
    for (unsigned k = 0; k < take; k++) st[k] = pi[8 * h + k];
    for (int k = 0; k < 12; k++) W(HIN + k, row) = st[k];
    for (int k = 0; k < 5; k++) W(HSWAP + k, row) = 0; /* no swap in a sponge row */
    permute_into_row(st, t, N, row);
  }
  gl_t pub[4] = {st[0], st[1], st[2], st[3]};
  /* the Merkle rows: the two-to-one compression of (left, right), the node on the side its position bit says */
  for (unsigned p = 0; p < n_paths; p++) {
    const gl_t* pw = paths + (size_t)p * path_words;
    uint64_t index = pw[0];
    gl_t node[4];
    for (int j = 0; j < 4; j++) node[j] = pi[path_pi0 + 8 * p + j];
    for (unsigned l = 0; l < depth; l++, index >>= 1) {
      const size_t row = MROW0 + p * depth + l;
      const gl_t* sib = pw + 1 + 4 * l;
      const int right = (int)(index & 1);
      gl_t m[12] = {0};
      for (int j = 0; j < 4; j++) {
        W(HIN + j, row) = node[j];
        W(HIN + 4 + j, row) = sib[j];
        W(HIN + 8 + j, row) = 0;
        W(HDELTA + j, row) = right ? gl_sub(sib[j], node[j]) : 0;
        m[j] = right ? sib[j] : node[j];
        m[4 + j] = right ? node[j] : sib[j];
      }
      W(HSWAP, row) = (gl_t)right;
      permute_into_row(m, t, N, row);
      for (int j = 0; j < 4; j++) node[j] = m[j];
    }
    /* the leaf sponge of the path: hash_no_pad of the opened row, eight words a row */
    if (leaf_len) {
      const gl_t* leaf = pw + 1 + 4 * depth;
      gl_t ls[12] = {0};
      for (unsigned h = 0; h < LH; h++) {
        const size_t row = LROW0 + p * LH + h;
        const unsigned take = leaf_len - 8 * h < 8 ? leaf_len - 8 * h : 8;
        for (unsigned k = 0; k < take; k++) ls[k] = leaf[8 * h + k];
        for (int k = 0; k < 12; k++) W(HIN + k, row) = ls[k];
        for (int k = 0; k < 5; k++) W(HSWAP + k, row) = 0;
        permute_into_row(ls, t, N, row);
      }
    }
  }
  for (int j = 0; j < 4; j++) W(j, 0) = pub[j];
  for (size_t i = AROW0; i < N; i++) {
    const gl_t c0 = consts[2 * N + i], c1 = consts[3 * N + i];
    const int p = (int)(i % 4);
    if (p == 1)
      for (int s = 0; s < SLOTS; s++) {
        W(4 * s, i) = W(4 * s + 3, i - 1);
        W(4 * s + 1, i) = W(4 * ((s + 1) % SLOTS) + 3, i - 1);
        W(4 * s + 2, i) = W(4 * s + 2, i - 1);
      }
    if (p == 0 && i == AROW0)
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
          den = gl_mul(den, gl_add(gl_add(w, gl_mul(beta, consts[(size_t)(SIG0 + j) * N + i])), gamma));
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
