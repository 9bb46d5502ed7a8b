// verifier.cpp -- CPU verifier of one table / recursion-shaped proof (the acceptance check behind
// VerifierState::verify, plonky_block_proof_gen/src/verifier_state.rs:56-71; upstream
// verify_stark_proof_with_challenges + fri::verifier::verify_fri_proof).  Host code only: the
// reference's verifier is CPU too, and a light verifier must not need a GPU.
#include <string>
#include <thread>
#include <system_error>
#include <vector>
#include "prover.hpp"

namespace bpg {

namespace {

using gl::Ext;

void merkle_leaf_digest(const uint64_t* data, size_t len, uint64_t out[4]) {
  if (len <= 4) {  // Hasher::hash_or_noop
    std::memset(out, 0, 32);
    std::memcpy(out, data, len * 8);
  } else {
    hash_no_pad_host(data, len, out);
  }
}
// merkle_proofs::verify_merkle_proof_to_cap
bool merkle_verify(const uint64_t* leaf, size_t leaf_len, uint64_t index, const uint64_t* path, uint32_t depth,
                   const uint64_t* cap) {
  uint64_t cur[4];
  merkle_leaf_digest(leaf, leaf_len, cur);
  for (uint32_t l = 0; l < depth; l++, path += 4, index >>= 1) {
    uint64_t s[12] = {0};
    if (index & 1) {
      std::memcpy(s, path, 32);
      std::memcpy(s + 4, cur, 32);
    } else {
      std::memcpy(s, cur, 32);
      std::memcpy(s + 4, path, 32);
    }
    poseidon_host(s);
    std::memcpy(cur, s, 32);
  }
  return std::memcmp(cur, cap + 4 * index, 32) == 0;
}

Ext rd(const uint64_t* p, size_t i) { return Ext{p[2 * i], p[2 * i + 1]}; }
// The opened row at zeta / g*zeta as the AIR evaluators (air.hpp) read it.
struct OpenedRow {
  const uint64_t *cst_, *loc_, *nxt_, *aux_, *aux_nxt_;
  Ext x_;                // the opening point zeta
  const uint64_t* pub_;  // the table's public inputs
  Ext x() const { return x_; }
  uint64_t pub(uint32_t j) const { return pub_[j]; }
  Ext cst(uint32_t k) const { return rd(cst_, k); }
  Ext loc(uint32_t c) const { return rd(loc_, c); }
  Ext nxt(uint32_t c) const { return rd(nxt_, c); }
  Ext aux(uint32_t k) const { return rd(aux_, k); }
  Ext aux_nxt(uint32_t k) const { return rd(aux_nxt_, k); }
};
// starky ConstraintConsumer over the extension field with base-field alphas: acc_j = sum_i c_i alpha_j^(T-1-i),
// which is acc = acc * alpha + c over the list, whatever order the evaluators emit in.
struct Consumer {
  std::vector<uint64_t> apow[2];  // alpha_j^e, e < T
  uint32_t T;
  Ext z_last, l_first, l_last;
  Ext acc[2];
  Consumer(uint32_t T_, uint64_t a0, uint64_t a1) : T(T_) {
    const uint64_t al[2] = {a0, a1};
    for (int j = 0; j < 2; j++) {
      apow[j].resize(T);
      uint64_t p = 1;
      for (uint32_t e = 0; e < T; e++, p = gl::mulc(p, al[j])) apow[j][e] = p;
      acc[j] = gl::ext(0);
    }
  }
  void all(uint32_t idx, Ext c) {
    for (int j = 0; j < 2; j++) acc[j] = gl::add(acc[j], gl::scale(c, apow[j][T - 1 - idx]));
  }
  void transition(uint32_t idx, Ext c) { all(idx, gl::mul(c, z_last)); }
  void first(uint32_t idx, Ext c) { all(idx, gl::mul(c, l_first)); }
  void last(uint32_t idx, Ext c) { all(idx, gl::mul(c, l_last)); }
};

// fri::verifier::compute_evaluation: interpolate the arity coset values, evaluate at beta.
Ext compute_evaluation(uint64_t x, uint32_t in_coset_br, uint32_t arity_bits, const uint64_t* evals, Ext beta) {
  const uint32_t arity = 1u << arity_bits;
  const uint64_t g = gl::root(arity_bits);
  const uint64_t start = gl::mulc(x, gl::pow(gl::inv(g), gl::bitrev(in_coset_br, arity_bits)));
  uint64_t pts[16];
  pts[0] = start;
  for (uint32_t j = 1; j < arity; j++) pts[j] = gl::mulc(pts[j - 1], g);
  Ext acc = gl::ext(0);
  for (uint32_t j = 0; j < arity; j++) {
    Ext num = gl::ext(1);
    uint64_t den = 1;
    for (uint32_t k = 0; k < arity; k++) {
      if (k == j) continue;
      num = gl::mul(num, gl::sub(beta, gl::ext(pts[k])));
      den = gl::mulc(den, gl::subc(pts[j], pts[k]));
    }
    acc = gl::add(acc, gl::mul(gl::scale(num, gl::inv(den)), rd(evals, gl::bitrev(j, arity_bits))));
  }
  return acc;
}

}  // namespace

#define REJECT(...) return fail(BP_ERR_VERIFY, __VA_ARGS__)

// The caller has driven `ch` through the prologue (circuit digest / caps observed, ctl drawn).
int stark_verify(const StarkCfg& cfg, const uint64_t* const_cap, const Ctl& ctl, Challenger& ch,
                 const uint64_t* P, size_t n_words) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  const ProofLayout L = proof_layout(cfg);
  if (n_words != L.total) REJECT("proof has %zu words, expected %zu", n_words, L.total);
  const uint32_t log_n = cfg.log_n, r = cfg.rate_bits, h = cfg.cap_height, log_m = log_n + r;
  const uint64_t N = (uint64_t)1 << log_n, M = N << r;
  const uint32_t C = cfg.n_cols, K = cfg.n_const, A = L.n_aux, Q = L.n_quot, qdf = 1u << r, arity = 1u << cfg.arity_bits;
  if (P[0] != PROOF_MAGIC || P[1] != log_n || P[2] != C || P[3] != K || P[6] != r || P[8] != cfg.num_queries ||
      P[9] != L.n_layers || P[10] != L.final_len || P[14] != cfg.air_id)
    REJECT("proof header does not match the circuit shape");
  for (size_t i = PROOF_HDR_WORDS; i < L.queries; i++)
    if (P[i] >= gl::P) REJECT("non-canonical field element at word %zu", i);
  if (K && !const_cap) return fail(BP_ERR_INVALID_INPUT, "constants cap missing");

  ch.observe(P + L.aux_cap, L.cap_words);
  const uint64_t alpha0 = ch.challenge(), alpha1 = ch.challenge();
  ch.observe(P + L.quot_cap, L.cap_words);
  const Ext zeta = ch.challenge_ext();
  const uint64_t *oz = P + L.open_zeta, *on = P + L.open_next, *of = P + L.open_first;

  {  // constraint check at zeta
    const uint64_t g = gl::root(log_n);
    const Ext zn = gl::pow(zeta, N), zh = gl::sub(zn, gl::ext(1));
    if (gl::eq(zh, gl::ext(0))) REJECT("Opening point is in the subgroup.");
    const Ext zhn = gl::scale(zh, gl::inv(N));
    const air::Shape shape{cfg.air_id, C, K, cfg.deg_pow};
    const uint32_t n_air = air::n_constraints(shape);
    Consumer k(n_air + air::ctl::n_constraints(shape), alpha0, alpha1);
    k.z_last = gl::sub(zeta, gl::ext(gl::inv(g)));
    k.l_first = gl::mul(zhn, gl::inv(gl::sub(zeta, gl::ext(1))));
    k.l_last = gl::mul(zhn, gl::inv(gl::sub(gl::scale(zeta, g), gl::ext(1))));
    const OpenedRow row{oz, oz + 2 * (size_t)K, on, oz + 2 * (size_t)(K + C), on + 2 * (size_t)C, zeta, ctl.pub};
    for (uint32_t u = 0; u < air::n_units(shape); u++) air::eval_unit<Ext>(shape, u, n_air, ctl.v, row, k);
    air::ctl::eval<Ext>(shape, n_air, 0, A, ctl.v, row, k);
    const uint64_t* oq = oz + 2 * (size_t)(K + C + A);
    for (int j = 0; j < 2; j++) {
      Ext acc = gl::ext(0);
      for (uint32_t t = qdf; t-- > 0;) acc = gl::add(gl::mul(acc, zn), rd(oq, j * qdf + t));
      if (!gl::eq(gl::mul(acc, zh), k.acc[j])) REJECT("constraint check at zeta failed (challenge %d)", j);
    }
  }
  ch.observe(oz, 2 * (size_t)L.n_zeta);
  ch.observe(on, 2 * (size_t)L.n_next);
  ch.observe(of, 2 * (size_t)A);

  const Ext alpha = ch.challenge_ext();
  Ext betas[8];
  for (uint32_t l = 0; l < L.n_layers; l++) {
    ch.observe(P + L.fri_caps + l * L.cap_words, L.cap_words);
    betas[l] = ch.challenge_ext();
  }
  const uint64_t* fp = P + L.final_poly;
  ch.observe(fp, 2 * (size_t)L.final_len);
  ch.observe(P[L.pow]);
  const uint64_t resp = ch.challenge();
  if (cfg.pow_bits && (resp >> (64 - cfg.pow_bits)) != 0) REJECT("proof-of-work check failed");

  const Ext points[3] = {zeta, gl::scale(zeta, gl::root(log_n)), gl::ext(1)};
  const uint64_t* opens[3] = {oz, on, of};
  const uint32_t kk[3] = {L.n_zeta, L.n_next, A};
  Ext red_open[3], a_shift[3];
  for (int b = 0; b < 3; b++) {
    Ext acc = gl::ext(0);
    for (size_t j = kk[b]; j-- > 0;) acc = gl::add(gl::mul(acc, alpha), rd(opens[b], j));
    red_open[b] = acc;
    a_shift[b] = gl::pow(alpha, kk[b]);
  }
  const uint64_t* caps[4];
  uint32_t widths[4];
  int n_init = 0;
  if (K) { caps[n_init] = const_cap; widths[n_init++] = K; }
  caps[n_init] = P + L.trace_cap; widths[n_init++] = C;
  caps[n_init] = P + L.aux_cap; widths[n_init++] = A;
  caps[n_init] = P + L.quot_cap; widths[n_init++] = Q;
  const uint64_t w_m = gl::root(log_m);

  // the query indices come out of the transcript one after the other; the queries themselves are independent
  std::vector<uint64_t> xs(cfg.num_queries);
  for (uint32_t q = 0; q < cfg.num_queries; q++) xs[q] = ch.challenge() & (M - 1);
  auto verify_query = [&](uint32_t q) -> int {
    const uint64_t* w = P + L.queries + (size_t)q * L.query_words;
    uint64_t x = xs[q];
    if (*w++ != x) REJECT("query %u: index does not match the transcript", q);
    const uint64_t* rows[4];
    for (int o = 0; o < n_init; o++) {
      rows[o] = w;
      w += widths[o];
      for (uint32_t c = 0; c < widths[o]; c++) if (rows[o][c] >= gl::P) REJECT("query %u: non-canonical element", q);
      if (!merkle_verify(rows[o], widths[o], x, w, L.depth0, caps[o])) REJECT("query %u: Merkle path of oracle %d", q, o);
      w += L.depth0 * 4;
    }
    const uint64_t sx = gl::mulc(gl::GENERATOR, gl::pow(w_m, gl::bitrev((uint32_t)x, log_m)));
    const uint64_t *cst = K ? rows[0] : nullptr, *tr = rows[K ? 1 : 0], *ax = rows[K ? 2 : 1], *qu = rows[K ? 3 : 2];
    Ext sum = gl::ext(0);
    for (int b = 0; b < 3; b++) {  // fri_combine_initial
      Ext acc = gl::ext(0);
      if (b == 0) for (size_t j = Q; j-- > 0;) acc = gl::add(gl::mul(acc, alpha), gl::ext(qu[j]));
      for (size_t j = A; j-- > 0;) acc = gl::add(gl::mul(acc, alpha), gl::ext(ax[j]));
      if (b < 2) for (size_t j = C; j-- > 0;) acc = gl::add(gl::mul(acc, alpha), gl::ext(tr[j]));
      if (b == 0) for (size_t j = K; j-- > 0;) acc = gl::add(gl::mul(acc, alpha), gl::ext(cst[j]));
      const Ext num = gl::sub(acc, red_open[b]), den = gl::sub(gl::ext(sx), points[b]);
      sum = gl::add(gl::mul(sum, a_shift[b]), gl::mul(num, gl::inv(den)));
    }
    Ext old_eval = sum;
    uint32_t lm = log_m;
    uint64_t subgroup_x = sx;
    for (uint32_t l = 0; l < L.n_layers; l++) {
      const uint64_t* evals = w;
      w += 2 * arity;
      for (uint32_t c = 0; c < 2 * arity; c++) if (evals[c] >= gl::P) REJECT("query %u: non-canonical element", q);
      const uint32_t in_coset = (uint32_t)(x & (arity - 1));
      const uint64_t leaf = x >> cfg.arity_bits;
      if (!gl::eq(rd(evals, in_coset), old_eval)) REJECT("query %u: FRI layer %u consistency", q, l);
      const uint32_t depth = lm - cfg.arity_bits - h;
      if (!merkle_verify(evals, 2 * arity, leaf, w, depth, P + L.fri_caps + l * L.cap_words))
        REJECT("query %u: Merkle path of FRI layer %u", q, l);
      w += depth * 4;
      old_eval = compute_evaluation(subgroup_x, in_coset, cfg.arity_bits, evals, betas[l]);
      subgroup_x = gl::pow(subgroup_x, arity);
      x = leaf;
      lm -= cfg.arity_bits;
    }
    Ext fv = gl::ext(0);
    for (size_t i = L.final_len; i-- > 0;) fv = gl::add(gl::scale(fv, subgroup_x), rd(fp, i));
    if (!gl::eq(fv, old_eval)) REJECT("query %u: final polynomial evaluation", q);
    return BP_OK;
  };
  // A proof of the default shape is ~3,000 host permutations of Merkle paths over its 28 / 84 queries (5 .. 10 ms): when
  // the process is not busy proving (a verifier, the top of a multi-GPU tree checking a foreign child) the queries are
  // checked by up to four threads; the reported failure is that of the lowest failing query either way.
  const uint32_t hw = std::thread::hardware_concurrency();
  const uint32_t n_threads = (cfg.num_queries >= 16 && hw >= 4 && provers_active() <= 1) ? 4 : 1;
  if (n_threads == 1) {
    for (uint32_t q = 0; q < cfg.num_queries; q++)
      if (int r = verify_query(q)) return r;
    return BP_OK;
  }
  std::vector<int> rcs(cfg.num_queries, BP_OK);
  std::vector<std::string> msgs(cfg.num_queries);
  auto work = [&](uint32_t t) {
    for (uint32_t q = t; q < cfg.num_queries; q += n_threads) {
      try {
        rcs[q] = verify_query(q);
        if (rcs[q]) msgs[q] = bp_last_error();
      } catch (...) {
        rcs[q] = BP_ERR_DEVICE;
        msgs[q] = "out of memory while verifying a query";
      }
    }
  };
  {
    std::vector<std::thread> pool;
    struct Join {
      std::vector<std::thread>& p;
      ~Join() { for (auto& t : p) if (t.joinable()) t.join(); }
    } join{pool};
    try {
      for (uint32_t t = 1; t < n_threads; t++) pool.emplace_back(work, t);
    } catch (const std::system_error&) {  // no threads to be had: the slices not started are done below
    }
    const uint32_t started = (uint32_t)pool.size() + 1;
    work(0);
    for (uint32_t t = started; t < n_threads; t++) work(t);
  }
  for (uint32_t q = 0; q < cfg.num_queries; q++)
    if (rcs[q]) return fail(rcs[q], "%s", msgs[q].c_str());
  return BP_OK;
}

}  // namespace bpg
