/* oracle/stark.c -- synthetic-AIR STARK prover + verifier (CPU restatement).
 * TEST INFRASTRUCTURE ONLY; "parity unpinned" by the reference (see gl.h).
 *
 * Flow restated from the upstream call reached at plonky_block_proof_gen/src/proof_gen.rs:44-52
 * (prove_root -> plonky2_evm prover::prove_single_table -> PolynomialBatch::prove_openings ->
 * fri_proof), SURVEY.md section 3.2.  The AIR itself is synthetic because
 * the zkEVM tables are upstream-only.  FRI here works in COEFFICIENT space exactly as upstream
 * does (reduce_polys_base, divide_by_linear, coset_fft per layer); the HIP path folds in the
 * evaluation domain, so agreement between the two is a real cross-check, not a tautology.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define MAGIC 0x4B52415453475042ULL /* "BPGSTARK" */
#define HDR_WORDS 16

struct orc_committed {
  unsigned log_n, rate_bits, cap_h;
  size_t n_cols;
  gl_t* coeffs;  /* [n_cols][N] natural order */
  gl_t* lde;     /* [n_cols][M] natural order, point i = 7*w_M^i */
  gl_t* digests; /* level order, leaf k = row bitrev(k) */
};

static void* xmalloc(size_t n) {
  void* p = malloc(n ? n : 1);
  if (!p) { fprintf(stderr, "oracle: out of memory (%zu bytes)\n", n); abort(); }
  return p;
}

/* ------------------------------------------------------------------ shape helpers */
uint32_t orc_cfg_n_aux(const orc_stark_cfg* c) { return orc_ctl_n_aux(c->air_id, c->n_cols); } /* ctl.c */
uint32_t orc_cfg_n_quot(const orc_stark_cfg* c) { return 2u << c->rate_bits; } /* 2 challenges x qdf */
/* FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits) */
uint32_t orc_cfg_n_layers(const orc_stark_cfg* c) {
  uint32_t d = c->log_n, n = 0;
  while (d > c->final_poly_bits && d + c->rate_bits >= c->cap_height + c->arity_bits) {
    if (d < c->arity_bits) break;
    d -= c->arity_bits; n++;
  }
  return n;
}
typedef struct {
  size_t cap_words, trace_cap, aux_cap, quot_cap, open_zeta, open_next, open_first, fri_caps,
      final_poly, pow, queries, query_words, total;
  uint32_t n_aux, n_quot, n_layers, final_len, n_zeta, n_next, depth0;
} layout_t;
static layout_t layout(const orc_stark_cfg* c) {
  layout_t L;
  memset(&L, 0, sizeof(L));
  L.n_aux = orc_cfg_n_aux(c); L.n_quot = orc_cfg_n_quot(c); L.n_layers = orc_cfg_n_layers(c);
  L.final_len = 1u << (c->log_n - L.n_layers * c->arity_bits);
  L.cap_words = ((size_t)4) << c->cap_height;
  L.n_zeta = c->n_const + c->n_cols + L.n_aux + L.n_quot;
  L.n_next = c->n_cols + L.n_aux;
  L.depth0 = c->log_n + c->rate_bits - c->cap_height;
  size_t o = HDR_WORDS;
  L.trace_cap = o; o += L.cap_words;
  L.aux_cap = o; o += L.cap_words;
  L.quot_cap = o; o += L.cap_words;
  L.open_zeta = o; o += 2 * (size_t)L.n_zeta;
  L.open_next = o; o += 2 * (size_t)L.n_next;
  L.open_first = o; o += 2 * (size_t)L.n_aux;
  L.fri_caps = o; o += L.cap_words * L.n_layers;
  L.final_poly = o; o += 2 * (size_t)L.final_len;
  L.pow = o; o += 1;
  L.queries = o;
  size_t q = 1; /* x_index */
  size_t n_oracle_cols = (size_t)c->n_const + c->n_cols + L.n_aux + L.n_quot;
  size_t n_oracles = c->n_const ? 4 : 3;
  q += n_oracle_cols + n_oracles * L.depth0 * 4;
  uint32_t lm = c->log_n + c->rate_bits;
  for (uint32_t l = 0; l < L.n_layers; l++) {
    q += (2u << c->arity_bits) + (size_t)(lm - c->arity_bits - c->cap_height) * 4;
    lm -= c->arity_bits;
  }
  L.query_words = q;
  L.total = o + q * c->num_queries;
  return L;
}
size_t orc_proof_words(const orc_stark_cfg* c) { return layout(c).total; }

/* ------------------------------------------------------------------ synthetic witness */
static inline uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static inline gl_t rnd(uint64_t seed, uint64_t col, uint64_t row) {
  return gl_canon(splitmix64(seed ^ (col << 32) ^ row));
}
static inline gl_t pow_e(gl_t t, uint32_t e) { return e == 3 ? gl_mul(gl_sqr(t), t) : t; }

void orc_synth_constants(uint64_t seed, unsigned log_n, size_t n_const, gl_t* consts) {
  size_t n = (size_t)1 << log_n;
  for (size_t k = 0; k < n_const; k++)
    for (size_t i = 0; i < n; i++) consts[k * n + i] = rnd(seed ^ 0xC0115700C0115700ULL, k, i);
}
/* Witness of the synthetic AIR: per group of 4 columns (a,b,c,d):
 *   c = a*b + q*a;  d[0] = a[0]+b[0];  d[i+1] = (a*b*c)[i]^e + b[i];  q = const col (g mod K) or 1. */
void orc_synth_trace(uint64_t seed, const orc_stark_cfg* cf, const gl_t* consts, gl_t* t) {
  size_t n = (size_t)1 << cf->log_n, C = cf->n_cols, G = C / 4;
#pragma omp parallel for schedule(static)
  for (size_t g = 0; g < G; g++) {
    gl_t *a = t + (4 * g) * n, *b = a + n, *c = b + n, *d = c + n;
    const gl_t* q = cf->n_const ? consts + (g % cf->n_const) * n : NULL;
    for (size_t i = 0; i < n; i++) {
      a[i] = rnd(seed, 4 * g, i);
      b[i] = rnd(seed, 4 * g + 1, i);
      gl_t ab = gl_mul(a[i], b[i]);
      c[i] = gl_add(ab, q ? gl_mul(q[i], a[i]) : a[i]);
    }
    d[0] = gl_add(a[0], b[0]);
    for (size_t i = 0; i + 1 < n; i++)
      d[i + 1] = gl_add(pow_e(gl_mul(gl_mul(a[i], b[i]), c[i]), cf->deg_pow), b[i]);
  }
  for (size_t c = 4 * G; c < C; c++)
    for (size_t i = 0; i < n; i++) t[c * n + i] = rnd(seed, c, i);
}

/* ------------------------------------------------------------------ commitments */
static orc_committed* commit_common(const gl_t* data, unsigned log_n, size_t n_cols, unsigned rate_bits,
                                    unsigned cap_h, int from_coeffs) {
  orc_committed* c = (orc_committed*)xmalloc(sizeof(*c));
  size_t n = (size_t)1 << log_n, m = n << rate_bits;
  c->log_n = log_n; c->rate_bits = rate_bits; c->cap_h = cap_h; c->n_cols = n_cols;
  c->coeffs = (gl_t*)xmalloc(n_cols * n * sizeof(gl_t));
  c->lde = (gl_t*)xmalloc(n_cols * m * sizeof(gl_t));
  c->digests = (gl_t*)xmalloc(orc_merkle_digest_words(log_n + rate_bits, cap_h) * sizeof(gl_t));
  orc_lde_batch(data, c->coeffs, c->lde, log_n, rate_bits, n_cols, from_coeffs);
  /* transpose + reverse_index_bits_in_place(leaves): leaf k = LDE row bitrev(k) */
  orc_merkle_commit(c->lde, m, n_cols, log_n + rate_bits, cap_h, 1, c->digests);
  return c;
}
orc_committed* orc_commit_values(const gl_t* v, unsigned log_n, size_t n_cols, unsigned r, unsigned h) {
  return commit_common(v, log_n, n_cols, r, h, 0);
}
orc_committed* orc_commit_coeffs(const gl_t* v, unsigned log_n, size_t n_cols, unsigned r, unsigned h) {
  return commit_common(v, log_n, n_cols, r, h, 1);
}
const gl_t* orc_committed_cap(const orc_committed* c) {
  return c->digests + orc_merkle_digest_words(c->log_n + c->rate_bits, c->cap_h) - ((size_t)4 << c->cap_h);
}
const gl_t* orc_committed_lde(const orc_committed* c) { return c->lde; }
const gl_t* orc_committed_coeffs(const orc_committed* c) { return c->coeffs; }
const gl_t* orc_committed_digests(const orc_committed* c) { return c->digests; }
void orc_committed_free(orc_committed* c) {
  if (!c) return;
  free(c->coeffs); free(c->lde); free(c->digests); free(c);
}

/* ------------------------------------------------------------------ AIR evaluation */
/* Constraint order (both prover and verifier): per group g: all-rows, transition, first-row;
 * then per aux column k: transition, last-row.  Consumer: acc_j = acc_j*alpha_j + constraint
 * (starky ConstraintConsumer::constraint).  Base-field version used on the LDE coset. */
typedef orc_consumer consumer_t;
#define cons orc_cons
static void eval_constraints_base(const orc_stark_cfg* cf, const gl_t* cst, const gl_t* loc,
                                  const gl_t* nxt, const gl_t* aux, const gl_t* aux_nxt,
                                  const gl_t ctl[4], gl_t x, consumer_t* k) {
  size_t G = cf->n_cols / 4, A = cf->n_cols / 8;
  if (cf->air_id == ORC_AIR_PLONK) { /* gates and copy constraints in one list (plonk_air.c) */
    orc_plonk_constraints_base(cst, loc, aux, aux_nxt, ctl, cf->pub, x, k);
    return;
  }
  if (cf->air_id == ORC_AIR_KECCAK_F) { /* the AIR's own list (keccak_air.c), then the table's lookups (ctl.c) */
    orc_keccak_constraints_base(loc, nxt, k);
    G = 0;
  } else if (cf->air_id == ORC_AIR_LOGIC) {
    orc_logic_constraints_base(loc, nxt, k);
    G = 0;
  } else if (cf->air_id == ORC_AIR_MEMORY) {
    orc_memory_constraints_base(loc, nxt, k);
    G = 0;
  } else if (cf->air_id == ORC_AIR_ARITHMETIC) {
    orc_arithmetic_constraints_base(loc, nxt, k);
    G = 0;
  } else if (cf->air_id == ORC_AIR_BYTE_PACKING) {
    orc_byte_packing_constraints_base(loc, nxt, k);
    G = 0;
  } else if (cf->air_id == ORC_AIR_KECCAK_SPONGE) {
    orc_keccak_sponge_constraints_base(loc, nxt, k);
    G = 0;
  } else if (cf->air_id == ORC_AIR_ARITHMETIC_MUL) {
    orc_arithmetic_mul_constraints_base(loc, nxt, k);
    G = 0;
  }
  for (size_t g = 0; g < G; g++) {
    gl_t a = loc[4 * g], b = loc[4 * g + 1], c = loc[4 * g + 2], d = loc[4 * g + 3];
    gl_t q = cf->n_const ? cst[g % cf->n_const] : 1;
    gl_t ab = gl_mul(a, b);
    cons(k, gl_sub(gl_sub(c, ab), gl_mul(q, a)));
    gl_t t = pow_e(gl_mul(ab, c), cf->deg_pow);
    cons(k, gl_mul(gl_sub(gl_sub(nxt[4 * g + 3], t), b), k->z_last));
    cons(k, gl_mul(gl_sub(gl_sub(d, a), b), k->l_first));
  }
  if (cf->air_id != ORC_AIR_SYNTHETIC) { orc_ctl_constraints_base(cf->air_id, loc, aux, aux_nxt, ctl, k); return; }
  for (size_t j = 0; j < A; j++) {
    gl_t beta = ctl[2 * (j & 1)], gamma = ctl[2 * (j & 1) + 1];
    gl_t term = gl_add(gl_add(gamma, loc[8 * j]), gl_mul(beta, loc[8 * j + 1]));
    cons(k, gl_mul(gl_sub(aux[j], gl_mul(aux_nxt[j], term)), k->z_last));
    cons(k, gl_mul(gl_sub(aux[j], term), k->l_last));
  }
}
/* Extension-field version for the verifier's check at zeta (alphas stay in the base field). */
typedef orc_consumer2 consumer2_t;
#define cons2 orc_cons2
static inline gl2_t pow_e2(gl2_t t, uint32_t e) { return e == 3 ? gl2_mul(gl2_sqr(t), t) : t; }
static void eval_constraints_ext(const orc_stark_cfg* cf, const gl2_t* cst, const gl2_t* loc,
                                 const gl2_t* nxt, const gl2_t* aux, const gl2_t* aux_nxt,
                                 const gl_t ctl[4], gl2_t x, consumer2_t* k) {
  size_t G = cf->n_cols / 4, A = cf->n_cols / 8;
  if (cf->air_id == ORC_AIR_PLONK) {
    orc_plonk_constraints_ext(cst, loc, aux, aux_nxt, ctl, cf->pub, x, k);
    return;
  }
  if (cf->air_id == ORC_AIR_KECCAK_F) {
    orc_keccak_constraints_ext(loc, nxt, k);
    G = 0;
  } else if (cf->air_id == ORC_AIR_LOGIC) {
    orc_logic_constraints_ext(loc, nxt, k);
    G = 0;
  } else if (cf->air_id == ORC_AIR_MEMORY) {
    orc_memory_constraints_ext(loc, nxt, k);
    G = 0;
  } else if (cf->air_id == ORC_AIR_ARITHMETIC) {
    orc_arithmetic_constraints_ext(loc, nxt, k);
    G = 0;
  } else if (cf->air_id == ORC_AIR_BYTE_PACKING) {
    orc_byte_packing_constraints_ext(loc, nxt, k);
    G = 0;
  } else if (cf->air_id == ORC_AIR_KECCAK_SPONGE) {
    orc_keccak_sponge_constraints_ext(loc, nxt, k);
    G = 0;
  } else if (cf->air_id == ORC_AIR_ARITHMETIC_MUL) {
    orc_arithmetic_mul_constraints_ext(loc, nxt, k);
    G = 0;
  }
  for (size_t g = 0; g < G; g++) {
    gl2_t a = loc[4 * g], b = loc[4 * g + 1], c = loc[4 * g + 2], d = loc[4 * g + 3];
    gl2_t q = cf->n_const ? cst[g % cf->n_const] : gl2_from(1);
    gl2_t ab = gl2_mul(a, b);
    cons2(k, gl2_sub(gl2_sub(c, ab), gl2_mul(q, a)));
    gl2_t t = pow_e2(gl2_mul(ab, c), cf->deg_pow);
    cons2(k, gl2_mul(gl2_sub(gl2_sub(nxt[4 * g + 3], t), b), k->z_last));
    cons2(k, gl2_mul(gl2_sub(gl2_sub(d, a), b), k->l_first));
  }
  if (cf->air_id != ORC_AIR_SYNTHETIC) { orc_ctl_constraints_ext(cf->air_id, loc, aux, aux_nxt, ctl, k); return; }
  for (size_t j = 0; j < A; j++) {
    gl_t beta = ctl[2 * (j & 1)], gamma = ctl[2 * (j & 1) + 1];
    gl2_t term = gl2_add(gl2_add(gl2_from(gamma), loc[8 * j]), gl2_scale(loc[8 * j + 1], beta));
    cons2(k, gl2_mul(gl2_sub(aux[j], gl2_mul(aux_nxt[j], term)), k->z_last));
    cons2(k, gl2_mul(gl2_sub(aux[j], term), k->l_last));
  }
}

/* ------------------------------------------------------------------ helpers */
static gl2_t eval_poly_base(const gl_t* coeffs, size_t n, gl2_t x) { /* Horner, PolynomialCoeffs::eval */
  gl2_t acc = gl2_from(0);
  for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, x), gl2_from(coeffs[i]));
  return acc;
}
static gl2_t eval_poly_ext(const gl2_t* coeffs, size_t n, gl2_t x) {
  gl2_t acc = gl2_from(0);
  for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, x), coeffs[i]);
  return acc;
}
static void observe_ext(orc_challenger* ch, const gl2_t* e, size_t n) {
  for (size_t i = 0; i < n; i++) { orc_ch_observe(ch, e[i].c0); orc_ch_observe(ch, e[i].c1); }
}
/* ext-coefficient polynomial -> values on shift*<w_m> (natural order): two base-field NTTs. */
static void coset_ntt_ext(gl2_t* a, unsigned log_m, gl_t shift) {
  size_t m = (size_t)1 << log_m;
  gl_t* t = (gl_t*)xmalloc(2 * m * sizeof(gl_t));
  for (size_t i = 0; i < m; i++) { t[i] = a[i].c0; t[m + i] = a[i].c1; }
  orc_coset_ntt(t, log_m, shift); orc_coset_ntt(t + m, log_m, shift);
  for (size_t i = 0; i < m; i++) a[i] = gl2(t[i], t[m + i]);
  free(t);
}

typedef struct { const orc_committed* o; size_t first, count; } poly_range;
typedef struct { gl2_t point; poly_range r[4]; int n_r; size_t k; } batch_t;

/* fri::prover::fri_proof_of_work, with the SMALLEST valid witness (upstream takes any). */
static gl_t pow_grind(const orc_challenger* ch, unsigned bits) {
  gl_t base[12];
  memcpy(base, ch->state, sizeof(base));
  for (unsigned i = 0; i < ch->n_in; i++) base[i] = ch->in[i];
  unsigned pos = ch->n_in; /* < 8 by the challenger invariant */
  for (uint64_t cand = 0;; cand += 4096) {
    uint64_t best = UINT64_MAX;
#pragma omp parallel for reduction(min : best)
    for (uint64_t k = 0; k < 4096; k++) {
      gl_t s[12];
      memcpy(s, base, sizeof(s));
      s[pos] = cand + k;
      orc_poseidon(s);
      if ((s[7] >> (64 - bits)) == 0 && cand + k < best) best = cand + k;
    }
    if (best != UINT64_MAX) return best;
  }
}

/* ------------------------------------------------------------------ prover */
/* Quotient values on the LDE coset (prover::compute_quotient_polys): for natural index i of the domain
 * 7*<w_M>, row i of the three LDE matrices (column stride M = n << rate_bits) and row i + 2^rate_bits
 * ("next"), all constraints folded with alpha0 / alpha1 and divided by Z_H.  qv: [2][M]. */
void orc_quotient_values(const orc_stark_cfg* cf, const gl_t* const_lde, const gl_t* trace_lde, const gl_t* aux_lde,
                         const gl_t ctl[4], gl_t alpha0, gl_t alpha1, gl_t* qv) {
  const unsigned log_n = cf->log_n, r = cf->rate_bits, log_m = log_n + r;
  const size_t N = (size_t)1 << log_n, M = N << r, C = cf->n_cols, K = cf->n_const, A = orc_cfg_n_aux(cf),
               qdf = (size_t)1 << r;
  {
    gl_t wM = gl_root(log_m), g = gl_root(log_n), ginv = gl_inv(g), ninv = gl_inv((gl_t)N);
    gl_t sN = gl_pow(GL_GENERATOR, N), wq = gl_root(r); /* x^N = 7^N * w_{2^r}^(i mod 2^r) */
#pragma omp parallel
    {
      gl_t* row = (gl_t*)xmalloc((K + 2 * C + 2 * A + 1) * sizeof(gl_t));
      gl_t *cst = row, *loc = cst + K, *nxt = loc + C, *ax = nxt + C, *axn = ax + A;
#pragma omp for schedule(static)
      for (size_t i = 0; i < M; i++) {
        size_t in = (i + qdf) & (M - 1);
        for (size_t c = 0; c < K; c++) cst[c] = const_lde[c * M + i];
        for (size_t c = 0; c < C; c++) { loc[c] = trace_lde[c * M + i]; nxt[c] = trace_lde[c * M + in]; }
        for (size_t c = 0; c < A; c++) { ax[c] = aux_lde[c * M + i]; axn[c] = aux_lde[c * M + in]; }
        gl_t x = gl_mul(GL_GENERATOR, gl_pow(wM, i));
        gl_t zh = gl_sub(gl_mul(sN, gl_pow(wq, i & (qdf - 1))), 1); /* x^N - 1, never 0 on the coset */
        consumer_t k;
        k.alpha[0] = alpha0; k.alpha[1] = alpha1; k.acc[0] = k.acc[1] = 0;
        k.z_last = gl_sub(x, ginv);
        k.l_first = gl_mul(gl_mul(zh, ninv), gl_inv(gl_sub(x, 1)));
        k.l_last = gl_mul(gl_mul(zh, ninv), gl_inv(gl_sub(gl_mul(g, x), 1)));
        eval_constraints_base(cf, cst, loc, nxt, ax, axn, ctl, x, &k);
        gl_t zhi = gl_inv(zh);
        qv[i] = gl_mul(k.acc[0], zhi);
        qv[M + i] = gl_mul(k.acc[1], zhi);
      }
      free(row);
    }
  }
}

int orc_stark_prove(const orc_stark_cfg* cf, const orc_committed* consts, const orc_committed* trace,
                    const gl_t* tv, const gl_t ctl[4], orc_challenger* ch, gl_t* proof) {
  layout_t L = layout(cf);
  const unsigned log_n = cf->log_n, r = cf->rate_bits, h = cf->cap_height, log_m = log_n + r;
  const size_t N = (size_t)1 << log_n, M = N << r, C = cf->n_cols, K = cf->n_const, A = L.n_aux,
               Q = L.n_quot, qdf = (size_t)1 << r;
  if ((cf->deg_pow != 1 && cf->deg_pow != 3) || qdf != 3 * cf->deg_pow - 1 || C < 8 || log_m < h ||
      (K && !consts) || cf->air_id > ORC_AIR_PLONK ||
      (cf->air_id == ORC_AIR_PLONK && (C != ORC_PLONK_COLS || K != ORC_PLONK_CONSTS || cf->deg_pow != 3)) ||
      (cf->air_id == ORC_AIR_ARITHMETIC_MUL && (C != ORC_ARITHMETIC_MUL_COLS || K != 0 || cf->deg_pow != 1)) ||
      (cf->air_id == ORC_AIR_KECCAK_SPONGE && (C != ORC_KECCAK_SPONGE_COLS || K != 0 || cf->deg_pow != 1)) ||
      (cf->air_id == ORC_AIR_BYTE_PACKING && (C != ORC_BYTE_PACKING_COLS || K != 0 || cf->deg_pow != 1)) ||
      (cf->air_id == ORC_AIR_ARITHMETIC && (C != ORC_ARITHMETIC_COLS || K != 0 || cf->deg_pow != 1)) ||
      (cf->air_id == ORC_AIR_MEMORY && (C != ORC_MEMORY_COLS || K != 0 || cf->deg_pow != 1)) ||
      (cf->air_id == ORC_AIR_KECCAK_F && (C != ORC_KECCAK_COLS || K != 0 || cf->deg_pow != 1)) ||
      (cf->air_id == ORC_AIR_LOGIC && (C != ORC_LOGIC_COLS || K != 0 || cf->deg_pow != 1)))
    return -1;
  memset(proof, 0, L.total * sizeof(gl_t));
  proof[0] = MAGIC; proof[1] = log_n; proof[2] = C; proof[3] = K; proof[4] = A; proof[5] = Q;
  proof[6] = r; proof[7] = h; proof[8] = cf->num_queries; proof[9] = L.n_layers;
  proof[10] = L.final_len; proof[11] = cf->deg_pow; proof[12] = cf->pow_bits; proof[13] = cf->arity_bits;
  proof[14] = cf->air_id;
  memcpy(proof + L.trace_cap, orc_committed_cap(trace), L.cap_words * 8);

  /* 1. auxiliary columns.  Synthetic table: suffix products of term = gamma + a + beta*b; a table with a real AIR:
   * its lookups (ctl.c) */
  gl_t* auxv = (gl_t*)xmalloc(A * N * sizeof(gl_t));
  if (cf->air_id == ORC_AIR_PLONK) { /* the copy products need the sigmas on the trace domain: evaluate the constants */
    gl_t* cv = (gl_t*)xmalloc(K * N * sizeof(gl_t));
    memcpy(cv, consts->coeffs, K * N * sizeof(gl_t));
    for (size_t c = 0; c < K; c++) orc_ntt(cv + c * N, log_n);
    orc_plonk_aux_columns(tv, cv, log_n, ctl, auxv);
    free(cv);
  } else if (cf->air_id != ORC_AIR_SYNTHETIC) {
    orc_ctl_aux_columns(cf->air_id, tv, log_n, ctl, auxv);
  } else {
#pragma omp parallel for
    for (size_t k = 0; k < A; k++) {
      const gl_t *a = tv + (8 * k) * N, *b = a + N;
      gl_t beta = ctl[2 * (k & 1)], gamma = ctl[2 * (k & 1) + 1], *z = auxv + k * N;
      z[N - 1] = gl_add(gl_add(gamma, a[N - 1]), gl_mul(beta, b[N - 1]));
      for (size_t i = N - 1; i-- > 0;)
        z[i] = gl_mul(z[i + 1], gl_add(gl_add(gamma, a[i]), gl_mul(beta, b[i])));
    }
  }
  orc_committed* aux = orc_commit_values(auxv, log_n, A, r, h);
  free(auxv);
  memcpy(proof + L.aux_cap, orc_committed_cap(aux), L.cap_words * 8);
  orc_ch_observe_many(ch, orc_committed_cap(aux), L.cap_words);

  /* 2. alphas (base field: challenger.get_n_challenges(num_challenges)) */
  gl_t alpha0 = orc_ch_challenge(ch), alpha1 = orc_ch_challenge(ch);

  /* 3. quotient values on the LDE coset, prover::compute_quotient_polys */
  gl_t* qv = (gl_t*)xmalloc(2 * M * sizeof(gl_t));
  orc_quotient_values(cf, K ? consts->lde : NULL, trace->lde, aux->lde, ctl, alpha0, alpha1, qv);
  /* values on the coset -> coefficients (degree < qdf*N) -> qdf chunks of N per challenge */
  orc_coset_intt(qv, log_m, GL_GENERATOR);
  orc_coset_intt(qv + M, log_m, GL_GENERATOR);
  orc_committed* quot = orc_commit_coeffs(qv, log_n, Q, r, h); /* chunk t of challenge j = column j*qdf+t */
  free(qv);
  memcpy(proof + L.quot_cap, orc_committed_cap(quot), L.cap_words * 8);
  orc_ch_observe_many(ch, orc_committed_cap(quot), L.cap_words);

  /* 4. zeta */
  gl2_t zeta = orc_ch_challenge_ext(ch);
  gl2_t zeta_next = gl2_scale(zeta, gl_root(log_n));

  /* 5. openings (StarkOpeningSet::new), oracle order: constants, trace, aux, quotient */
  batch_t B[3];
  memset(B, 0, sizeof(B));
  int nr = 0;
  B[0].point = zeta;
  if (K) B[0].r[nr++] = (poly_range){consts, 0, K};
  B[0].r[nr++] = (poly_range){trace, 0, C};
  B[0].r[nr++] = (poly_range){aux, 0, A};
  B[0].r[nr++] = (poly_range){quot, 0, Q};
  B[0].n_r = nr; B[0].k = L.n_zeta;
  B[1].point = zeta_next; B[1].r[0] = (poly_range){trace, 0, C}; B[1].r[1] = (poly_range){aux, 0, A};
  B[1].n_r = 2; B[1].k = L.n_next;
  B[2].point = gl2_from(1); B[2].r[0] = (poly_range){aux, 0, A}; B[2].n_r = 1; B[2].k = A;
  size_t open_off[3] = {L.open_zeta, L.open_next, L.open_first};
  for (int b = 0; b < 3; b++) {
    gl2_t* out = (gl2_t*)(proof + open_off[b]);
    size_t idx = 0;
    for (int ri = 0; ri < B[b].n_r; ri++) {
      const orc_committed* o = B[b].r[ri].o;
      size_t cnt = B[b].r[ri].count;
#pragma omp parallel for
      for (size_t c = 0; c < cnt; c++) out[idx + c] = eval_poly_base(o->coeffs + c * N, N, B[b].point);
      idx += cnt;
    }
  }
  /* challenger.observe_openings(&openings.to_fri_openings()) */
  for (int b = 0; b < 3; b++) observe_ext(ch, (const gl2_t*)(proof + open_off[b]), B[b].k);

  /* 6. PolynomialBatch::prove_openings */
  gl2_t alpha = orc_ch_challenge_ext(ch);
  gl2_t* fin = (gl2_t*)xmalloc(M * sizeof(gl2_t));
  for (size_t i = 0; i < M; i++) fin[i] = gl2_from(0);
  for (int b = 0; b < 3; b++) {
    /* composition = sum_j alpha^j f_j  (ReducingFactor::reduce_polys_base: Horner from the back) */
    gl2_t* comp = (gl2_t*)xmalloc(N * sizeof(gl2_t));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < N; i++) {
      gl2_t acc = gl2_from(0);
      for (int ri = B[b].n_r - 1; ri >= 0; ri--) {
        const orc_committed* o = B[b].r[ri].o;
        for (size_t c = B[b].r[ri].count; c-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), gl2_from(o->coeffs[c * N + i]));
      }
      comp[i] = acc;
    }
    /* quotient = (comp - comp(z)) / (X - z): synthetic division; padded back with a zero */
    gl2_t* quo = (gl2_t*)xmalloc(N * sizeof(gl2_t));
    quo[N - 1] = gl2_from(0);
    gl2_t carry = gl2_from(0);
    for (size_t i = N; i-- > 1;) {
      carry = gl2_add(comp[i], gl2_mul(carry, B[b].point));
      quo[i - 1] = carry;
    }
    /* alpha.shift_poly(final): final *= alpha^(#polys in this batch); final += quotient */
    gl2_t ak = gl2_pow(alpha, B[b].k);
    for (size_t i = 0; i < N; i++) fin[i] = gl2_add(gl2_mul(fin[i], ak), quo[i]);
    free(comp); free(quo);
  }
  /* lde_final_poly / lde_final_values */
  gl2_t* coeffs = (gl2_t*)xmalloc(M * sizeof(gl2_t));
  gl2_t* values = (gl2_t*)xmalloc(M * sizeof(gl2_t));
  memcpy(coeffs, fin, M * sizeof(gl2_t));
  memcpy(values, fin, M * sizeof(gl2_t));
  free(fin);
  coset_ntt_ext(values, log_m, GL_GENERATOR);

  /* fri::prover::fri_committed_trees */
  gl_t** layer_digests = (gl_t**)xmalloc((L.n_layers + 1) * sizeof(gl_t*));
  gl2_t** layer_values = (gl2_t**)xmalloc((L.n_layers + 1) * sizeof(gl2_t*));
  unsigned lm = log_m;
  gl_t shift = GL_GENERATOR;
  size_t arity = (size_t)1 << cf->arity_bits;
  for (uint32_t l = 0; l < L.n_layers; l++) {
    size_t m = (size_t)1 << lm;
    gl2_t* br = (gl2_t*)xmalloc(m * sizeof(gl2_t)); /* reverse_index_bits_in_place(values) */
    for (size_t i = 0; i < m; i++) br[i] = values[bitrev32((uint32_t)i, lm)];
    layer_values[l] = br;
    unsigned log_leaves = lm - cf->arity_bits;
    gl_t* dg = (gl_t*)xmalloc(orc_merkle_digest_words(log_leaves, h) * sizeof(gl_t));
    orc_merkle_commit_rows((const gl_t*)br, 2 * arity, log_leaves, h, dg);
    layer_digests[l] = dg;
    const gl_t* cap = dg + orc_merkle_digest_words(log_leaves, h) - L.cap_words;
    memcpy(proof + L.fri_caps + l * L.cap_words, cap, L.cap_words * 8);
    orc_ch_observe_many(ch, cap, L.cap_words);
    gl2_t beta = orc_ch_challenge_ext(ch);
    /* coeffs.chunks(arity).map(reduce_with_powers(beta)) */
    size_t m2 = m >> cf->arity_bits;
    for (size_t c = 0; c < m2; c++) {
      gl2_t acc = gl2_from(0);
      for (size_t j = arity; j-- > 0;) acc = gl2_add(gl2_mul(acc, beta), coeffs[c * arity + j]);
      coeffs[c] = acc;
    }
    shift = gl_pow(shift, arity);
    lm -= cf->arity_bits;
    memcpy(values, coeffs, m2 * sizeof(gl2_t));
    coset_ntt_ext(values, lm, shift);
  }
  /* truncate the zero tail and send the final polynomial */
  memcpy(proof + L.final_poly, coeffs, L.final_len * sizeof(gl2_t));
  for (size_t i = L.final_len; i < ((size_t)1 << lm); i++)
    if (coeffs[i].c0 || coeffs[i].c1) { fprintf(stderr, "oracle: FRI final poly tail not zero\n"); return -2; }
  observe_ext(ch, coeffs, L.final_len);
  free(coeffs); free(values);

  /* proof of work */
  gl_t nonce = pow_grind(ch, cf->pow_bits);
  proof[L.pow] = nonce;
  orc_ch_observe(ch, nonce);
  gl_t resp = orc_ch_challenge(ch);
  if (cf->pow_bits && (resp >> (64 - cf->pow_bits)) != 0) return -3;

  /* fri_prover_query_rounds: indices first, then the openings */
  const orc_committed* initial[4];
  int n_init = 0;
  if (K) initial[n_init++] = consts;
  initial[n_init++] = trace; initial[n_init++] = aux; initial[n_init++] = quot;
  uint64_t* xs = (uint64_t*)xmalloc(cf->num_queries * sizeof(uint64_t));
  for (uint32_t q = 0; q < cf->num_queries; q++) xs[q] = orc_ch_challenge(ch) & (M - 1);
  for (uint32_t q = 0; q < cf->num_queries; q++) {
    gl_t* w = proof + L.queries + (size_t)q * L.query_words;
    size_t x = xs[q];
    *w++ = x;
    size_t row = bitrev32((uint32_t)x, log_m);
    for (int o = 0; o < n_init; o++) {
      for (size_t c = 0; c < initial[o]->n_cols; c++) *w++ = initial[o]->lde[c * M + row];
      orc_merkle_path(initial[o]->digests, log_m, h, x, w);
      w += L.depth0 * 4;
    }
    unsigned llm = log_m;
    for (uint32_t l = 0; l < L.n_layers; l++) {
      size_t leaf = x >> cf->arity_bits;
      memcpy(w, layer_values[l] + leaf * arity, arity * sizeof(gl2_t));
      w += 2 * arity;
      unsigned log_leaves = llm - cf->arity_bits;
      orc_merkle_path(layer_digests[l], log_leaves, h, leaf, w);
      w += (log_leaves - h) * 4;
      x = leaf;
      llm -= cf->arity_bits;
    }
  }
  free(xs);
  for (uint32_t l = 0; l < L.n_layers; l++) { free(layer_digests[l]); free(layer_values[l]); }
  free(layer_digests); free(layer_values);
  orc_committed_free(aux); orc_committed_free(quot);
  return 0;
}

/* ------------------------------------------------------------------ verifier */
/* fri::verifier::compute_evaluation: interpolate the coset values and evaluate at beta. */
static gl2_t compute_evaluation(gl_t x, size_t x_in_coset_br, unsigned arity_bits, const gl2_t* evals,
                                gl2_t beta) {
  size_t arity = (size_t)1 << arity_bits;
  gl_t g = gl_root(arity_bits);
  /* evals are in bit-reversed order; the coset starts at x * g^(-rev(x_in_coset)) */
  size_t rev = bitrev32((uint32_t)x_in_coset_br, arity_bits);
  gl_t start = gl_mul(x, gl_pow(gl_inv(g), rev));
  /* Lagrange interpolation over points start*g^j, values evals[bitrev(j)] */
  gl2_t acc = gl2_from(0);
  for (size_t j = 0; j < arity; j++) {
    gl_t xj = gl_mul(start, gl_pow(g, j));
    gl2_t num = gl2_from(1);
    gl_t den = 1;
    for (size_t k = 0; k < arity; k++) {
      if (k == j) continue;
      gl_t xk = gl_mul(start, gl_pow(g, k));
      num = gl2_mul(num, gl2_sub(beta, gl2_from(xk)));
      den = gl_mul(den, gl_sub(xj, xk));
    }
    acc = gl2_add(acc, gl2_mul(gl2_scale(num, gl_inv(den)), evals[bitrev32((uint32_t)j, arity_bits)]));
  }
  return acc;
}

int orc_stark_verify(const orc_stark_cfg* cf, const gl_t* const_cap, const gl_t ctl[4],
                     orc_challenger* ch, const gl_t* proof) {
  layout_t L = layout(cf);
  const unsigned log_n = cf->log_n, r = cf->rate_bits, h = cf->cap_height, log_m = log_n + r;
  const size_t N = (size_t)1 << log_n, M = N << r, C = cf->n_cols, K = cf->n_const, A = L.n_aux,
               Q = L.n_quot, qdf = (size_t)1 << r, arity = (size_t)1 << cf->arity_bits;
  if (proof[0] != MAGIC || proof[1] != log_n || proof[2] != C || proof[3] != K || proof[6] != r ||
      proof[8] != cf->num_queries || proof[9] != L.n_layers || proof[10] != L.final_len || proof[14] != cf->air_id)
    return -1;
  for (size_t i = HDR_WORDS; i < L.total; i++) if (i < L.queries && proof[i] >= GL_P) return -2;

  orc_ch_observe_many(ch, proof + L.aux_cap, L.cap_words);
  gl_t alpha0 = orc_ch_challenge(ch), alpha1 = orc_ch_challenge(ch);
  orc_ch_observe_many(ch, proof + L.quot_cap, L.cap_words);
  gl2_t zeta = orc_ch_challenge_ext(ch);
  const gl2_t* oz = (const gl2_t*)(proof + L.open_zeta);
  const gl2_t* on = (const gl2_t*)(proof + L.open_next);
  const gl2_t* of = (const gl2_t*)(proof + L.open_first);

  /* constraint check at zeta: sum_t zeta^(tN) q_{j,t}(zeta) * Z_H(zeta) == acc_j(zeta) */
  {
    gl_t g = gl_root(log_n);
    gl2_t zn = gl2_pow(zeta, N), zh = gl2_sub(zn, gl2_from(1));
    if (gl2_eq(zh, gl2_from(0))) return -3; /* "Opening point is in the subgroup." */
    gl2_t ninv = gl2_from(gl_inv((gl_t)N));
    consumer2_t k;
    k.alpha[0] = alpha0; k.alpha[1] = alpha1; k.acc[0] = k.acc[1] = gl2_from(0);
    k.z_last = gl2_sub(zeta, gl2_from(gl_inv(g)));
    k.l_first = gl2_mul(gl2_mul(zh, ninv), gl2_inv(gl2_sub(zeta, gl2_from(1))));
    k.l_last = gl2_mul(gl2_mul(zh, ninv), gl2_inv(gl2_sub(gl2_scale(zeta, g), gl2_from(1))));
    eval_constraints_ext(cf, oz, oz + K, on, oz + K + C, on + C, ctl, zeta, &k);
    for (int j = 0; j < 2; j++) {
      gl2_t acc = gl2_from(0);
      for (size_t t = qdf; t-- > 0;) acc = gl2_add(gl2_mul(acc, zn), oz[K + C + A + j * qdf + t]);
      if (!gl2_eq(gl2_mul(acc, zh), k.acc[j])) return -4;
    }
  }
  observe_ext(ch, oz, L.n_zeta); observe_ext(ch, on, L.n_next); observe_ext(ch, of, A);

  /* FRI challenges (fri::challenges) */
  gl2_t alpha = orc_ch_challenge_ext(ch);
  gl2_t betas[32];
  for (uint32_t l = 0; l < L.n_layers; l++) {
    orc_ch_observe_many(ch, proof + L.fri_caps + l * L.cap_words, L.cap_words);
    betas[l] = orc_ch_challenge_ext(ch);
  }
  const gl2_t* final_poly = (const gl2_t*)(proof + L.final_poly);
  observe_ext(ch, final_poly, L.final_len);
  orc_ch_observe(ch, proof[L.pow]);
  gl_t resp = orc_ch_challenge(ch);
  if (cf->pow_bits && (resp >> (64 - cf->pow_bits)) != 0) return -5;

  /* PrecomputedReducedOpenings: sum_j alpha^j opening_j per batch */
  gl2_t points[3] = {zeta, gl2_scale(zeta, gl_root(log_n)), gl2_from(1)};
  const gl2_t* opens[3] = {oz, on, of};
  size_t kk[3] = {L.n_zeta, L.n_next, A};
  gl2_t red_open[3];
  for (int b = 0; b < 3; b++) {
    gl2_t acc = gl2_from(0);
    for (size_t j = kk[b]; j-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), opens[b][j]);
    red_open[b] = acc;
  }
  const gl_t* caps[4];
  size_t widths[4];
  int n_init = 0;
  if (K) { caps[n_init] = const_cap; widths[n_init++] = K; }
  caps[n_init] = proof + L.trace_cap; widths[n_init++] = C;
  caps[n_init] = proof + L.aux_cap; widths[n_init++] = A;
  caps[n_init] = proof + L.quot_cap; widths[n_init++] = Q;

  for (uint32_t q = 0; q < cf->num_queries; q++) {
    const gl_t* w = proof + L.queries + (size_t)q * L.query_words;
    size_t x = orc_ch_challenge(ch) & (M - 1);
    if (*w++ != x) return -6;
    const gl_t* rows[4];
    for (int o = 0; o < n_init; o++) {
      rows[o] = w; w += widths[o];
      if (orc_merkle_verify(rows[o], widths[o], x, w, log_m, h, caps[o])) return -7;
      w += L.depth0 * 4;
    }
    /* fri_combine_initial */
    gl_t sx = gl_mul(GL_GENERATOR, gl_pow(gl_root(log_m), bitrev32((uint32_t)x, log_m)));
    const gl_t* cst = K ? rows[0] : NULL;
    const gl_t *tr = rows[K ? 1 : 0], *ax = rows[K ? 2 : 1], *qu = rows[K ? 3 : 2];
    gl2_t sum = gl2_from(0);
    for (int b = 0; b < 3; b++) {
      gl2_t acc = gl2_from(0); /* alpha.reduce(evals) Horner from the back, batch poly order */
      if (b == 0) {
        for (size_t j = Q; j-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), gl2_from(qu[j]));
        for (size_t j = A; j-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), gl2_from(ax[j]));
        for (size_t j = C; j-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), gl2_from(tr[j]));
        for (size_t j = K; j-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), gl2_from(cst[j]));
      } else if (b == 1) {
        for (size_t j = A; j-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), gl2_from(ax[j]));
        for (size_t j = C; j-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), gl2_from(tr[j]));
      } else {
        for (size_t j = A; j-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), gl2_from(ax[j]));
      }
      gl2_t num = gl2_sub(acc, red_open[b]);
      gl2_t den = gl2_sub(gl2_from(sx), points[b]);
      sum = gl2_add(gl2_mul(sum, gl2_pow(alpha, kk[b])), gl2_mul(num, gl2_inv(den)));
    }
    gl2_t old_eval = sum;
    unsigned lm = log_m;
    gl_t subgroup_x = sx;
    for (uint32_t l = 0; l < L.n_layers; l++) {
      const gl2_t* evals = (const gl2_t*)w;
      w += 2 * arity;
      size_t in_coset = x & (arity - 1), leaf = x >> cf->arity_bits;
      if (!gl2_eq(evals[in_coset], old_eval)) return -8;
      unsigned log_leaves = lm - cf->arity_bits;
      if (orc_merkle_verify((const gl_t*)evals, 2 * arity, leaf, w, log_leaves, h,
                            proof + L.fri_caps + l * L.cap_words))
        return -9;
      w += (log_leaves - h) * 4;
      old_eval = compute_evaluation(subgroup_x, in_coset, cf->arity_bits, evals, betas[l]);
      subgroup_x = gl_pow(subgroup_x, arity);
      x = leaf;
      lm -= cf->arity_bits;
    }
    if (!gl2_eq(eval_poly_ext(final_poly, L.final_len, gl2_from(subgroup_x)), old_eval)) return -10;
  }
  return 0;
}

/* What an aggregating circuit walks for a child proof: the first query's opening of the trace oracle.  leaf = the digest
 * of the opened row (hash_or_noop), cap_entry = the entry of the trace cap above it, path = the leaf's position below that
 * entry followed by the 4 * depth0 sibling words (1 + 4 * depth0 words). */
void orc_proof_first_query_path(const orc_stark_cfg* cf, const gl_t* proof, gl_t leaf[4], gl_t cap_entry[4], gl_t* path) {
  layout_t L = layout(cf);
  const gl_t* q = proof + L.queries;
  const uint64_t x = q[0];
  const gl_t* row = q + 1 + (cf->n_const ? cf->n_const + (size_t)L.depth0 * 4 : 0); /* past the constants' row and path */
  if (cf->n_cols <= 4) {
    memset(leaf, 0, 32);
    memcpy(leaf, row, cf->n_cols * 8);
  } else {
    orc_hash_no_pad(row, cf->n_cols, leaf);
  }
  path[0] = x % ((uint64_t)1 << L.depth0);
  memcpy(path + 1, row + cf->n_cols, (size_t)L.depth0 * 32);
  memcpy(cap_entry, proof + L.trace_cap + 4 * (x >> L.depth0), 32);
}

void orc_proof_first_query_row(const orc_stark_cfg* cf, const gl_t* proof, gl_t* row_out) {
  layout_t L = layout(cf);
  const gl_t* q = proof + L.queries;
  memcpy(row_out, q + 1 + (cf->n_const ? cf->n_const + (size_t)L.depth0 * 4 : 0), cf->n_cols * sizeof(gl_t));
}

void orc_proof_digest(const orc_stark_cfg* cf, const gl_t* proof, gl_t out[4]) {
  layout_t L = layout(cf);
  size_t n = 3 * L.cap_words + 2 * (size_t)L.final_len + 1;
  gl_t* buf = (gl_t*)xmalloc(n * sizeof(gl_t));
  memcpy(buf, proof + L.trace_cap, 3 * L.cap_words * 8);
  memcpy(buf + 3 * L.cap_words, proof + L.final_poly, 2 * (size_t)L.final_len * 8);
  buf[n - 1] = proof[L.pow];
  orc_hash_no_pad(buf, n, out);
  free(buf);
}
