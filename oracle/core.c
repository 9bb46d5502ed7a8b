/* oracle/core.c -- NTT, Poseidon, Merkle, challenger, FRI fold.  TEST INFRASTRUCTURE ONLY.
 * "parity unpinned" by the reference (see gl.h).  Each function names the upstream algorithm it
 * restates; the only in-tree anchors are the call sites proof_gen.rs:44-52/:66-75/:97-103.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ NTT (K2) */
/* plonky2_field::fft: values[i] = sum_j coeffs[j] * w^(i*j), w = gl_root(log_n). */
void orc_dft_naive(const gl_t* in, gl_t* out, unsigned log_n, int inverse) {
  size_t n = (size_t)1 << log_n;
  gl_t w = gl_root(log_n);
  if (inverse) w = gl_inv(w);
  gl_t ninv = inverse ? gl_inv((gl_t)n % GL_P) : 1;
  for (size_t i = 0; i < n; i++) {
    gl_t wi = gl_pow(w, i), x = 1, acc = 0;
    for (size_t j = 0; j < n; j++) { acc = gl_add(acc, gl_mul(in[j], x)); x = gl_mul(x, wi); }
    out[i] = gl_mul(acc, ninv);
  }
}

static void ntt_core(gl_t* a, unsigned log_n, gl_t w) {
  size_t n = (size_t)1 << log_n;
  for (size_t i = 0; i < n; i++) { /* bit-reverse, then decimation-in-time */
    size_t j = bitrev32((uint32_t)i, log_n);
    if (i < j) { gl_t t = a[i]; a[i] = a[j]; a[j] = t; }
  }
  for (unsigned s = 1; s <= log_n; s++) {
    size_t m = (size_t)1 << s, half = m >> 1;
    gl_t wm = w;
    for (unsigned k = s; k < log_n; k++) wm = gl_sqr(wm);
    for (size_t k = 0; k < n; k += m) {
      gl_t x = 1;
      for (size_t j = 0; j < half; j++) {
        gl_t u = a[k + j], v = gl_mul(a[k + j + half], x);
        a[k + j] = gl_add(u, v);
        a[k + j + half] = gl_sub(u, v);
        x = gl_mul(x, wm);
      }
    }
  }
}
void orc_ntt(gl_t* a, unsigned log_n) { ntt_core(a, log_n, gl_root(log_n)); }
void orc_intt(gl_t* a, unsigned log_n) {
  size_t n = (size_t)1 << log_n;
  ntt_core(a, log_n, gl_inv(gl_root(log_n)));
  gl_t ninv = gl_inv((gl_t)n);
  for (size_t i = 0; i < n; i++) a[i] = gl_mul(a[i], ninv);
}
/* PolynomialCoeffs::coset_fft: scale coeff j by shift^j, then fft. */
void orc_coset_ntt(gl_t* a, unsigned log_n, gl_t shift) {
  size_t n = (size_t)1 << log_n;
  gl_t x = 1;
  for (size_t i = 0; i < n; i++) { a[i] = gl_mul(a[i], x); x = gl_mul(x, shift); }
  orc_ntt(a, log_n);
}
void orc_coset_intt(gl_t* a, unsigned log_n, gl_t shift) {
  size_t n = (size_t)1 << log_n;
  orc_intt(a, log_n);
  gl_t si = gl_inv(shift), x = 1;
  for (size_t i = 0; i < n; i++) { a[i] = gl_mul(a[i], x); x = gl_mul(x, si); }
}
void orc_ntt_batch(gl_t* cols, unsigned log_n, size_t n_cols, size_t stride, int inverse) {
#pragma omp parallel for schedule(dynamic)
  for (size_t c = 0; c < n_cols; c++) {
    if (inverse) orc_intt(cols + c * stride, log_n); else orc_ntt(cols + c * stride, log_n);
  }
}
/* PolynomialBatch::from_values / from_coeffs: ifft, zero-pad to n*2^r, coset_fft(shift = 7). */
void orc_lde_batch(const gl_t* values, gl_t* coeffs_out, gl_t* lde_out, unsigned log_n,
                   unsigned rate_bits, size_t n_cols, int from_coeffs) {
  size_t n = (size_t)1 << log_n, m = n << rate_bits;
#pragma omp parallel for schedule(dynamic)
  for (size_t c = 0; c < n_cols; c++) {
    gl_t* out = lde_out + c * m;
    memcpy(out, values + c * n, n * sizeof(gl_t));
    if (!from_coeffs) orc_intt(out, log_n);
    if (coeffs_out) memcpy(coeffs_out + c * n, out, n * sizeof(gl_t));
    memset(out + n, 0, (m - n) * sizeof(gl_t));
    orc_coset_ntt(out, log_n + rate_bits, GL_GENERATOR);
  }
}

/* ------------------------------------------------------------------ Poseidon (K3) */
/* plonky2::hash::poseidon, width 12, x^7, 4 full + 22 partial + 4 full rounds; the naive round
 * structure (poseidon_naive upstream) -- the optimised partial-round form upstream is
 * algebraically identical.  Constants: tools/gen_poseidon_constants.py (KAT-checked). */
static const gl_t POSEIDON_RC[360] = {
#include "poseidon_rc.inc"
};
static const uint64_t MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const uint64_t MDS_DIAG0 = 8;

static inline gl_t sbox7(gl_t x) {
  gl_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x2, x);
  return gl_mul(x3, x4);
}
static inline void mds_layer(gl_t s[12]) {
  /* entries < 2^6: accumulate the 32-bit halves separately in 64 bits (no overflow), recombine once */
  uint64_t lo[24], hi[24], o[12];
  for (int i = 0; i < 12; i++) { lo[i] = lo[i + 12] = (uint32_t)s[i]; hi[i] = hi[i + 12] = s[i] >> 32; }
  for (int r = 0; r < 12; r++) {
    uint64_t L = 0, H = 0;
    for (int i = 0; i < 12; i++) { L += lo[i + r] * MDS_CIRC[i]; H += hi[i + r] * MDS_CIRC[i]; }
    if (r == 0) { L += lo[0] * MDS_DIAG0; H += hi[0] * MDS_DIAG0; }
    o[r] = gl_reduce128((u128)L + ((u128)H << 32));
  }
  memcpy(s, o, sizeof(o));
}
void orc_poseidon(gl_t s[12]) {
  int rnd = 0;
  for (int phase = 0; phase < 3; phase++) {
    int n_rounds = phase == 1 ? 22 : 4;
    for (int k = 0; k < n_rounds; k++, rnd++) {
      for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], POSEIDON_RC[rnd * 12 + i]);
      if (phase == 1) s[0] = sbox7(s[0]);
      else for (int i = 0; i < 12; i++) s[i] = sbox7(s[i]);
      mds_layer(s);
    }
  }
}
void orc_poseidon_batch(gl_t* states, size_t n) {
#pragma omp parallel for
  for (size_t i = 0; i < n; i++) orc_poseidon(states + 12 * i);
}
/* hashing::hash_n_to_m_no_pad: overwrite-mode sponge, rate 8, no padding, digest = state[0..4]. */
void orc_hash_no_pad(const gl_t* in, size_t len, gl_t out[4]) {
  gl_t s[12] = {0};
  for (size_t off = 0; off < len; off += 8) {
    size_t k = len - off < 8 ? len - off : 8;
    memcpy(s, in + off, k * sizeof(gl_t));
    orc_poseidon(s);
  }
  memcpy(out, s, 4 * sizeof(gl_t));
}
/* Hasher::hash_or_noop: inputs of <= 4 elements are the digest, zero padded. */
void orc_hash_or_noop(const gl_t* in, size_t len, gl_t out[4]) {
  if (len <= 4) {
    memset(out, 0, 4 * sizeof(gl_t));
    memcpy(out, in, len * sizeof(gl_t));
  } else {
    orc_hash_no_pad(in, len, out);
  }
}
/* PoseidonHash::two_to_one: state = [l, r, 0,0,0,0], permute, first 4. */
void orc_two_to_one(const gl_t l[4], const gl_t r[4], gl_t out[4]) {
  gl_t s[12] = {0};
  memcpy(s, l, 32); memcpy(s + 4, r, 32);
  orc_poseidon(s);
  memcpy(out, s, 32);
}

/* ------------------------------------------------------------------ Merkle (K4) */
/* plonky2::hash::merkle_tree::MerkleTree::new(leaves, cap_height).  Our digest buffer is plain
 * level order (leaves first); cap[j] is the root over leaves [j*L/2^h, (j+1)*L/2^h). */
size_t orc_merkle_digest_words(unsigned log_leaves, unsigned cap_h) {
  return (((size_t)2 << log_leaves) - ((size_t)1 << cap_h)) * 4;
}
static void merkle_upper(gl_t* digests, unsigned log_leaves, unsigned cap_h) {
  gl_t* lvl = digests;
  for (unsigned l = log_leaves; l > cap_h; l--) {
    size_t cnt = (size_t)1 << l;
    gl_t* nxt = lvl + cnt * 4;
#pragma omp parallel for if (cnt > 1024)
    for (size_t i = 0; i < cnt / 2; i++) orc_two_to_one(lvl + 8 * i, lvl + 8 * i + 4, nxt + 4 * i);
    lvl = nxt;
  }
}
void orc_merkle_commit(const gl_t* cols, size_t stride, size_t n_cols, unsigned log_leaves,
                       unsigned cap_h, int bitrev_rows, gl_t* digests_out) {
  size_t L = (size_t)1 << log_leaves;
#pragma omp parallel
  {
    gl_t* row = (gl_t*)malloc((n_cols ? n_cols : 1) * sizeof(gl_t));
#pragma omp for
    for (size_t k = 0; k < L; k++) {
      size_t r = bitrev_rows ? bitrev32((uint32_t)k, log_leaves) : k;
      for (size_t c = 0; c < n_cols; c++) row[c] = cols[c * stride + r];
      orc_hash_or_noop(row, n_cols, digests_out + 4 * k);
    }
    free(row);
  }
  merkle_upper(digests_out, log_leaves, cap_h);
}
void orc_merkle_commit_rows(const gl_t* leaves, size_t leaf_len, unsigned log_leaves, unsigned cap_h,
                            gl_t* digests_out) {
  size_t L = (size_t)1 << log_leaves;
#pragma omp parallel for
  for (size_t k = 0; k < L; k++) orc_hash_or_noop(leaves + k * leaf_len, leaf_len, digests_out + 4 * k);
  merkle_upper(digests_out, log_leaves, cap_h);
}
/* MerkleTree::prove: siblings bottom-up, stopping below the cap. */
void orc_merkle_path(const gl_t* digests, unsigned log_leaves, unsigned cap_h, size_t leaf,
                     gl_t* path_out) {
  const gl_t* lvl = digests;
  size_t idx = leaf;
  for (unsigned l = log_leaves; l > cap_h; l--) {
    memcpy(path_out, lvl + 4 * (idx ^ 1), 32);
    path_out += 4;
    lvl += ((size_t)1 << l) * 4;
    idx >>= 1;
  }
}
/* merkle_proofs::verify_merkle_proof_to_cap */
int orc_merkle_verify(const gl_t* leaf_data, size_t leaf_len, size_t leaf, const gl_t* path,
                      unsigned log_leaves, unsigned cap_h, const gl_t* cap) {
  gl_t cur[4], nxt[4];
  orc_hash_or_noop(leaf_data, leaf_len, cur);
  size_t idx = leaf;
  for (unsigned l = log_leaves; l > cap_h; l--) {
    if (idx & 1) orc_two_to_one(path, cur, nxt); else orc_two_to_one(cur, path, nxt);
    memcpy(cur, nxt, 32);
    path += 4;
    idx >>= 1;
  }
  return memcmp(cur, cap + 4 * idx, 32) == 0 ? 0 : -1;
}

/* ------------------------------------------------------------------ Challenger (K7) */
/* plonky2::iop::challenger::Challenger: overwrite-mode duplex, outputs popped from the back. */
void orc_ch_init(orc_challenger* c) { memset(c, 0, sizeof(*c)); }
static void ch_duplex(orc_challenger* c) {
  for (unsigned i = 0; i < c->n_in; i++) c->state[i] = c->in[i];
  c->n_in = 0;
  orc_poseidon(c->state);
  memcpy(c->out, c->state, 8 * sizeof(gl_t));
  c->n_out = 8;
}
void orc_ch_observe(orc_challenger* c, gl_t e) {
  c->n_out = 0;
  c->in[c->n_in++] = e;
  if (c->n_in == 8) ch_duplex(c);
}
void orc_ch_observe_many(orc_challenger* c, const gl_t* e, size_t n) {
  for (size_t i = 0; i < n; i++) orc_ch_observe(c, e[i]);
}
gl_t orc_ch_challenge(orc_challenger* c) {
  if (c->n_in || !c->n_out) ch_duplex(c);
  return c->out[--c->n_out];
}
gl2_t orc_ch_challenge_ext(orc_challenger* c) {
  gl_t a = orc_ch_challenge(c), b = orc_ch_challenge(c);
  return gl2(a, b);
}

/* ------------------------------------------------------------------ FRI fold (K6b) */
/* Evaluation-domain statement of fri::prover::fri_committed_trees' fold
 * (coeffs.chunks(arity).map(reduce_with_powers(beta)) followed by coset_fft(shift^arity)).
 * For chunk c of the bit-reversed layer: the `arity` values are P(x*w_a^j) in bitrev(j) order with
 * x = shift * w_m^bitrev(c); interpolate u_i = x^i P_i(y) by an inverse DFT and evaluate
 * sum_i (beta/x)^i u_i  (fri::verifier::compute_evaluation does the same per query). */
void orc_fri_fold(const gl2_t* in, gl2_t* out, unsigned log_m, unsigned arity_bits, gl_t shift,
                  gl2_t beta) {
  size_t arity = (size_t)1 << arity_bits, n_out = ((size_t)1 << log_m) >> arity_bits;
  gl_t wm = gl_root(log_m), wa_inv = gl_inv(gl_root(arity_bits));
  gl_t ainv = gl_inv((gl_t)arity), shift_inv = gl_inv(shift);
#pragma omp parallel for if (n_out > 256)
  for (size_t c = 0; c < n_out; c++) {
    gl_t xinv = gl_mul(shift_inv, gl_inv(gl_pow(wm, bitrev32((uint32_t)c, log_m - arity_bits))));
    gl2_t bx = gl2_scale(beta, xinv), bxi = gl2_from(1), acc = gl2_from(0);
    for (size_t i = 0; i < arity; i++) {
      gl2_t u = gl2_from(0);
      for (size_t j = 0; j < arity; j++) { /* u_i = 1/a * sum_j w_a^(-ij) P(x w_a^j) */
        gl2_t v = in[c * arity + bitrev32((uint32_t)j, arity_bits)];
        u = gl2_add(u, gl2_scale(v, gl_pow(wa_inv, (i * j) % arity)));
      }
      acc = gl2_add(acc, gl2_mul(gl2_scale(u, ainv), bxi));
      bxi = gl2_mul(bxi, bx);
    }
    out[c] = acc;
  }
}
