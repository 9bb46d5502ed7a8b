// prover.cpp -- host orchestration of one table proof (prove_single_table + prove_openings +
// fri_proof of upstream plonky2_evm / plonky2 @ 265d46a9, reached from
// plonky_block_proof_gen/src/proof_gen.rs:44-52).  All field-heavy work is launched on the
// worker's HIP stream; the host keeps only the Fiat-Shamir transcript (strictly sequential
// Poseidon duplexing, K7) and the byte assembly of the proof.  Every host<->device hand-off is a
// challenge boundary of the protocol: caps and openings come down, challenges go up as kernel
// arguments.
#include <cstdio>
#include <cstdlib>
#include "prover.hpp"
#include <algorithm>

namespace bpg {

// ------------------------------------------------------------------ host Poseidon
static const uint64_t RC_HOST[360] = {
#include "poseidon_rc.inc"
};
static inline uint64_t sbox_host(uint64_t x) {
  uint64_t x2 = gl::mulc(x, x), x4 = gl::mulc(x2, x2), x3 = gl::mulc(x2, x);
  return gl::mulc(x3, x4);
}
void poseidon_host(uint64_t s[12]) {
  static const uint64_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  int rnd = 0;
  for (int phase = 0; phase < 3; phase++) {
    const int n_rounds = phase == 1 ? 22 : 4;
    for (int k = 0; k < n_rounds; k++, rnd++) {
      for (int i = 0; i < 12; i++) s[i] = gl::addc(gl::canon(s[i]), RC_HOST[rnd * 12 + i]);
      if (phase == 1) s[0] = sbox_host(s[0]);
      else for (int i = 0; i < 12; i++) s[i] = sbox_host(s[i]);
      uint64_t o[12];
      for (int r = 0; r < 12; r++) {
        unsigned __int128 acc = 0;
        for (int i = 0; i < 12; i++) acc += (unsigned __int128)s[(i + r) % 12] * C[i];
        if (r == 0) acc += (unsigned __int128)s[0] * 8;
        o[r] = gl::canon(gl::reduce128((uint64_t)acc, (uint64_t)(acc >> 64)));
      }
      std::memcpy(s, o, sizeof(o));
    }
  }
}
void hash_no_pad_host(const uint64_t* in, size_t len, uint64_t out[4]) {
  uint64_t s[12] = {0};
  for (size_t off = 0; off < len; off += 8) {
    size_t k = std::min<size_t>(8, len - off);
    std::memcpy(s, in + off, k * 8);
    poseidon_host(s);
  }
  std::memcpy(out, s, 32);
}

// ------------------------------------------------------------------ shapes
static uint32_t n_fri_layers(const StarkCfg& c) {  // FriReductionStrategy::ConstantArityBits
  uint32_t d = c.log_n, n = 0;
  while (d > c.final_poly_bits && d + c.rate_bits >= c.cap_height + c.arity_bits && d >= c.arity_bits) {
    d -= c.arity_bits;
    n++;
  }
  return n;
}
std::atomic<int> g_k5_spread_all{0};  // measurement knob (bp_tune_k5_spread): the loaded-device spreading rule for the synthetic AIR too
int check_cfg(const StarkCfg& c) {
  const air::Info* ai = air::info(c.air_id);
  if (!ai) return fail(BP_ERR_INVALID_INPUT, "unknown air_id %u (bp_air_count() AIRs are built in)", c.air_id);
  if (ai->n_cols && (c.n_cols != ai->n_cols || c.n_const != 0 || c.deg_pow != 1))
    return fail(BP_ERR_INVALID_INPUT, "AIR %u (%s) has %u columns, no constant columns and degree %u (deg_pow 1): got n_cols=%u "
                "n_const=%u deg_pow=%u", c.air_id, ai->name, ai->n_cols, ai->degree, c.n_cols, c.n_const, c.deg_pow);
  if (c.deg_pow != 1 && c.deg_pow != 3) return fail(BP_ERR_INVALID_INPUT, "deg_pow must be 1 or 3");
  if ((1u << c.rate_bits) != 3 * c.deg_pow - 1)
    return fail(BP_ERR_INVALID_INPUT, "quotient degree factor 3*deg_pow-1 must equal 2^rate_bits");
  if (c.n_cols < 8 || c.n_cols > 65536) return fail(BP_ERR_INVALID_INPUT, "n_cols out of range");
  if (c.log_n < 4 || c.log_n + c.rate_bits > 30 || c.log_n + c.rate_bits < c.cap_height)
    return fail(BP_ERR_INVALID_INPUT, "log_n out of range");
  if (c.arity_bits != 4) return fail(BP_ERR_UNSUPPORTED, "only arity_bits = 4 is built");
  if (c.num_queries == 0 || c.num_queries > 128) return fail(BP_ERR_INVALID_INPUT, "num_queries out of range");
  if (c.pow_bits > 32 || c.cap_height > 8) return fail(BP_ERR_INVALID_INPUT, "pow_bits/cap_height out of range");
  // the final polynomial is interpolated on the host in O(len^2): at most 256 points by design
  if (c.final_poly_bits > 8) return fail(BP_ERR_INVALID_INPUT, "final_poly_bits must be <= 8");
  if (c.n_const > 4096) return fail(BP_ERR_INVALID_INPUT, "n_const out of range");
  if (n_fri_layers(c) > 8) return fail(BP_ERR_INVALID_INPUT, "too many FRI layers");
  return BP_OK;
}
ProofLayout proof_layout(const StarkCfg& c) {
  ProofLayout L{};
  L.n_aux = c.n_cols / 8;
  L.n_quot = 2u << c.rate_bits;
  L.n_layers = n_fri_layers(c);
  L.final_len = 1u << (c.log_n - L.n_layers * c.arity_bits);
  L.cap_words = (size_t)4 << c.cap_height;
  L.n_zeta = c.n_const + c.n_cols + L.n_aux + L.n_quot;
  L.n_next = c.n_cols + L.n_aux;
  L.depth0 = c.log_n + c.rate_bits - c.cap_height;
  size_t o = PROOF_HDR_WORDS;
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
  size_t q = 1 + (size_t)L.n_zeta + (size_t)(c.n_const ? 4 : 3) * L.depth0 * 4;
  uint32_t lm = c.log_n + c.rate_bits;
  for (uint32_t l = 0; l < L.n_layers; l++) {
    q += (2u << c.arity_bits) + (size_t)(lm - c.arity_bits - c.cap_height) * 4;
    lm -= c.arity_bits;
  }
  L.query_words = q;
  L.total = o + q * c.num_queries;
  return L;
}
void proof_digest(const StarkCfg& c, const uint64_t* proof, uint64_t out[4]) {
  ProofLayout L = proof_layout(c);
  std::vector<uint64_t> buf;
  buf.insert(buf.end(), proof + L.trace_cap, proof + L.trace_cap + 3 * L.cap_words);
  buf.insert(buf.end(), proof + L.final_poly, proof + L.final_poly + 2 * (size_t)L.final_len);
  buf.push_back(proof[L.pow]);
  hash_no_pad_host(buf.data(), buf.size(), out);
}

// ------------------------------------------------------------------ arena / worker
int DeviceArena::init(size_t bytes) {
  destroy();
  BPG_HIP(hipMalloc(reinterpret_cast<void**>(&base_), bytes));
  cap_ = bytes;
  off_ = high_ = 0;
  return BP_OK;
}
void DeviceArena::destroy() {
  if (base_) (void)hipFree(base_);
  base_ = nullptr;
  cap_ = off_ = 0;
}
uint64_t* DeviceArena::alloc_words(size_t words) {
  size_t bytes = (words * 8 + 255) & ~(size_t)255;
  if (off_ + bytes > cap_) return nullptr;
  char* p = base_ + off_;
  off_ += bytes;
  high_ = std::max(high_, off_);
  return reinterpret_cast<uint64_t*>(p);
}
int Worker::init(int dev, size_t arena_bytes) {
  device = dev;
  // The device's host-wait mode is chosen BEFORE this library makes its first stream on it and never changed after:
  // a worker whose stream and event were made under spinning waits and that is still alive when the device is
  // switched to blocking waits makes a later hipFree (a device-wide wait) hang forever (tools/hang_probe.py; found
  // in round 3 by running the table-proof tests, which park a worker, before the first bp_state_build).
  (void)bp_use_blocking_sync(dev);
  BPG_HIP(hipSetDevice(dev));
  BPG_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  int rc = arena.init(arena_bytes);
  if (rc) return rc;
  pinned_words = (size_t)1 << 22;  // 32 MiB staging
  BPG_HIP(hipHostMalloc(reinterpret_cast<void**>(&pinned), pinned_words * 8, hipHostMallocDefault));
  BPG_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&pinned_dev), pinned, 0));
  BPG_HIP(hipMalloc(reinterpret_cast<void**>(&d_pow_result), 8));
  // many prover threads share few host cores: wait on a blocking event (the thread sleeps) rather
  // than spin in hipStreamSynchronize
  BPG_HIP(hipEventCreateWithFlags(&sync_event, hipEventBlockingSync | hipEventDisableTiming));
  return init_ntt_kernels();
}
void Worker::destroy() {
  if (stream) (void)hipStreamSynchronize(stream);
  arena.destroy();
  if (pinned) (void)hipHostFree(pinned);
  if (d_pow_result) (void)hipFree(d_pow_result);
  if (sync_event) (void)hipEventDestroy(sync_event);
  if (stream) (void)hipStreamDestroy(stream);
  sync_event = nullptr;
  pinned = nullptr;
  d_pow_result = nullptr;
  stream = nullptr;
}
int Worker::wait() {
  BPG_HIP(hipEventRecord(sync_event, stream));
  BPG_HIP(hipEventSynchronize(sync_event));
  return BP_OK;
}
int Worker::d2h(uint64_t* host_dst, const uint64_t* dev_src, size_t words) {
  for (size_t off = 0; off < words; off += pinned_words) {
    size_t k = std::min(pinned_words, words - off);
    BPG_HIP(hipMemcpyAsync(pinned, dev_src + off, k * 8, hipMemcpyDeviceToHost, stream));
    BPG_HIP(hipEventRecord(sync_event, stream));
    BPG_HIP(hipEventSynchronize(sync_event));
    std::memcpy(host_dst + off, pinned, k * 8);
  }
  return BP_OK;
}

#define ARENA_ALLOC(var, words)                                                                         \
  uint64_t* var = w.arena.alloc_words(words);                                                           \
  if (!var)                                                                                             \
    return fail(BP_ERR_DEVICE, "device arena exhausted (%zu MiB) allocating %zu words for " #var,     \
                w.arena.capacity() >> 20, (size_t)(words))
#define TRY(expr)        \
  do {                   \
    int _rc = (expr);    \
    if (_rc) return _rc; \
  } while (0)

extern "C" int bp_lde_batch(const uint64_t*, uint64_t, uint64_t*, uint64_t, uint64_t*, uint64_t, uint32_t, uint32_t,
                            uint32_t, int, void*);
extern "C" uint64_t bp_merkle_digest_words(uint32_t, uint32_t);

int commit(Worker& w, const uint64_t* d_in, uint32_t n_cols, uint32_t log_n, uint32_t rate_bits,
           uint32_t cap_height, bool from_coeffs, Committed* out) {
  const uint64_t n = (uint64_t)1 << log_n, m = n << rate_bits;
  out->log_n = log_n; out->n_cols = n_cols; out->rate_bits = rate_bits; out->cap_height = cap_height;
  ARENA_ALLOC(lde, (size_t)n_cols * m);
  const size_t dw = bp_merkle_digest_words(log_n + rate_bits, cap_height);
  ARENA_ALLOC(digests, dw);
  uint64_t* coeffs = const_cast<uint64_t*>(d_in);
  if (!from_coeffs) {
    ARENA_ALLOC(c, (size_t)n_cols * n);
    coeffs = c;
  }
  TRY(bp_lde_batch(d_in, n, from_coeffs ? nullptr : coeffs, n, lde, m, log_n, rate_bits, n_cols, from_coeffs,
                   w.stream));
  // the kernel that makes the cap level writes it into the pinned mailbox as well: no copy launch
  bool mirrored = false;
  TRY(merkle_commit_cols(lde, m, n_cols, log_n, rate_bits, cap_height, digests, w.stream, w.pinned_dev, &mirrored));
  out->coeffs = coeffs; out->lde = lde; out->digests = digests;
  const size_t cw = (size_t)4 << cap_height;
  out->cap.resize(cw);
  if (!mirrored) return w.d2h(out->cap.data(), digests + dw - cw, cw);
  TRY(w.wait());
  std::memcpy(out->cap.data(), w.pinned, cw * 8);
  return BP_OK;
}

// Everything of the quotient launch that follows from the table shape and the challenges (the caller
// sets the three LDE pointers and the two output buffers).  Used by stark_prove and bp_quotient_eval.
int quotient_args(const StarkCfg& cfg, const Ctl& ctl, uint64_t alpha0, uint64_t alpha1, QuotArgs* out) {
  QuotArgs& qa = *out;
  const uint32_t log_n = cfg.log_n, r = cfg.rate_bits, R = 1u << r, C = cfg.n_cols, K = cfg.n_const, A = C / 8;
  const uint64_t N = (uint64_t)1 << log_n, M = N << r;
  const uint64_t wM = gl::root(log_n + r), wN = gl::root(log_n);
  qa.trace_stride = qa.aux_stride = qa.const_stride = M;
  TRY(get_table(0, log_n, 0, &qa.tw_n));
  qa.air_id = cfg.air_id;
  qa.log_n = log_n; qa.rate_bits = r; qa.n_cols = C; qa.n_const = K; qa.n_aux = A; qa.deg_pow = cfg.deg_pow;
  // the constraint list and its units (air.hpp)
  const air::Shape shape{cfg.air_id, C, K, cfg.deg_pow};
  qa.n_air_constraints = air::n_constraints(shape);
  qa.n_constraints = qa.n_air_constraints + 2 * A;
  qa.n_air_units = air::n_units(shape);
  qa.aux_per_unit = std::max<uint32_t>(16, (A + 15) / 16);
  qa.n_ctl_units = (A + qa.aux_per_unit - 1) / qa.aux_per_unit;
  // One pass (the alpha fold never leaves the registers) once the rows alone fill the chip: 2048 workgroups of
  // 256 lanes = 2 per SIMD.  Shorter tables spread their units over grid.y until the launch has that many.
  // While several provers share the device nothing needs filling: one pass, one launch fewer on the proof's
  // critical path and no partial sums -- for the synthetic AIR, whose row costs little.  The AIRs of the real tables
  // cost 10^4 .. 10^5 instructions per row: one pass over a SHORT table is then milliseconds of a few workgroups on the
  // critical path (Keccak sponge, 2^9 rows: 2.6 ms against 0.11 ms spread, profiles/r3_k5_air_probe.txt), so their
  // units are spread until the launch has 256 workgroups.
  const uint32_t n_units = qa.n_air_units + qa.n_ctl_units, wg_x = (uint32_t)((M + 255) / 256);
  const uint32_t loaded_rows = (cfg.air_id == air::SYNTHETIC && !g_k5_spread_all.load(std::memory_order_relaxed))
                                   ? 1 : std::min<uint32_t>(n_units, std::max<uint32_t>(1, (256 + wg_x - 1) / wg_x));
  const uint32_t want_rows = device_loaded() ? loaded_rows : std::min<uint32_t>(n_units, std::max<uint32_t>(1, (2048 + wg_x - 1) / wg_x));
  qa.units_per_wg = (n_units + want_rows - 1) / want_rows;
  qa.alpha0 = alpha0; qa.alpha1 = alpha1;
  qa.g = wN; qa.g_inv = gl::inv(wN); qa.n_inv = gl::inv(N);
  // per-coset constants: g_t = 7 * w_M^t, Z_H(g_t x) = g_t^n - 1 (constant on a coset)
  for (uint32_t t = 0; t < R; t++) {
    qa.g_t[t] = gl::mulc(gl::GENERATOR, gl::pow(wM, t));
    qa.zh_t[t] = gl::subc(gl::pow(qa.g_t[t], N), 1);
    qa.zh_inv_t[t] = gl::inv(qa.zh_t[t]);
  }
  qa.ctl = ctl;
  return BP_OK;
}
size_t quotient_partial_words(const QuotArgs& qa) {
  const uint32_t n_units = qa.n_air_units + qa.n_ctl_units, wg_rows = (n_units + qa.units_per_wg - 1) / qa.units_per_wg;
  return wg_rows > 1 ? (size_t)wg_rows * 2 * (((size_t)1 << qa.log_n) << qa.rate_bits) : 0;
}

// One FRI layer of n_l << r extension values on the domain shift * <w_{n_l 2^r}> (coset-major).
int fri_layer_args(uint32_t log_nl, uint32_t rate_bits, uint32_t arity_bits, uint64_t shift, FriLayerArgs* out) {
  FriLayerArgs& fa = *out;
  const uint32_t R = 1u << rate_bits, arity = 1u << arity_bits;
  if (arity_bits != 4) return fail(BP_ERR_UNSUPPORTED, "only arity_bits = 4 is built");
  if (log_nl < arity_bits || rate_bits > 4) return fail(BP_ERR_INVALID_INPUT, "FRI layer too small or rate too high");
  fa.log_nl = log_nl; fa.rate_bits = rate_bits; fa.arity_bits = arity_bits;
  TRY(get_table(1, log_nl, 0, &fa.tw_nl_inv));
  const uint64_t wMl = gl::root(log_nl + rate_bits), wa_inv = gl::inv(gl::root(arity_bits));
  for (uint32_t t = 0; t < R; t++) fa.g_t_inv[t] = gl::inv(gl::mulc(shift, gl::pow(wMl, t)));
  for (uint32_t k = 0; k < arity; k++) fa.wa_inv_pow[k] = gl::pow(wa_inv, k);
  fa.arity_inv = gl::inv(arity);
  return BP_OK;
}

// ------------------------------------------------------------------ one table
int stark_prove(Worker& w, const StarkCfg& cfg, const Committed* consts, const Committed& trace,
                const uint64_t* d_tv, const Ctl& ctl, Challenger& ch, std::vector<uint64_t>& proof) {
  TRY(check_cfg(cfg));
  const ProofLayout L = proof_layout(cfg);
  const uint32_t log_n = cfg.log_n, r = cfg.rate_bits, h = cfg.cap_height, R = 1u << r;
  const uint64_t N = (uint64_t)1 << log_n, M = N << r;
  const uint32_t C = cfg.n_cols, K = cfg.n_const, A = L.n_aux, Q = L.n_quot;
  if (K && !consts) return fail(BP_ERR_INVALID_INPUT, "constants commitment missing");
  hipStream_t st = w.stream;
  proof.assign(L.total, 0);
  uint64_t* P = proof.data();
  P[0] = PROOF_MAGIC; P[1] = log_n; P[2] = C; P[3] = K; P[4] = A; P[5] = Q; P[6] = r; P[7] = h;
  P[8] = cfg.num_queries; P[9] = L.n_layers; P[10] = L.final_len; P[11] = cfg.deg_pow; P[12] = cfg.pow_bits;
  P[13] = cfg.arity_bits; P[14] = cfg.air_id;
  std::memcpy(P + L.trace_cap, trace.cap.data(), L.cap_words * 8);
  if (w.aborted()) return fail(BP_ERR_ABORTED, "aborted before auxiliary commitment");

  // per-coset constants: g_t = 7 * w_M^t, Z_H(g_t x) = g_t^n - 1 (constant on a coset)
  const uint64_t wM = gl::root(log_n + r), wN = gl::root(log_n);
  uint64_t g_t[16], zh_t[16], zh_inv_t[16];
  for (uint32_t t = 0; t < R; t++) {
    g_t[t] = gl::mulc(gl::GENERATOR, gl::pow(wM, t));
    zh_t[t] = gl::subc(gl::pow(g_t[t], N), 1);
    zh_inv_t[t] = gl::inv(zh_t[t]);
  }
  const uint64_t *tw_n = nullptr, *coset_scale = nullptr, *coset_scale_inv = nullptr;
  TRY(get_table(0, log_n, 0, &tw_n));
  TRY(get_table(2, log_n, r, &coset_scale));
  TRY(get_table(3, log_n, r, &coset_scale_inv));

  // 1. auxiliary columns (suffix products) and their commitment
  ARENA_ALLOC(d_auxv, (size_t)A * N);
  TRY(launch_aux(d_tv, d_auxv, log_n, A, ctl, st));
  Committed aux;
  TRY(commit(w, d_auxv, A, log_n, r, h, false, &aux));
  std::memcpy(P + L.aux_cap, aux.cap.data(), L.cap_words * 8);
  ch.observe(aux.cap.data(), L.cap_words);

  // 2. alphas
  const uint64_t alpha0 = ch.challenge(), alpha1 = ch.challenge();

  // 3. quotient on the LDE coset -> per-coset iNTT -> chunk coefficients -> commitment
  QuotArgs qa{};
  qa.trace_lde = trace.lde; qa.aux_lde = aux.lde; qa.const_lde = K ? consts->lde : nullptr;
  TRY(quotient_args(cfg, ctl, alpha0, alpha1, &qa));
  const size_t mark_q = w.arena.mark();
  ARENA_ALLOC(d_qc, (size_t)Q * N);  // chunk coefficients: live until the end (quotient oracle)
  const size_t mark_tmp = w.arena.mark();
  ARENA_ALLOC(d_cpow, 2 * (size_t)qa.n_constraints + 48);
  ARENA_ALLOC(d_partial, quotient_partial_words(qa) + 1);
  ARENA_ALLOC(d_qvals, 2 * M);
  qa.apow = d_cpow; qa.partial = d_partial; qa.qvals = d_qvals;
  TRY(launch_quotient(qa, st));
  TRY(intt_nat2br(d_qvals, N, d_qvals, N, log_n, 2 * R, true, st));  // 2 challenges x 2^r cosets, in place
  ChunkArgs ca{};
  ca.e = d_qvals; ca.inv_scale = coset_scale_inv; ca.out = d_qc; ca.out_stride = N; ca.log_n = log_n; ca.rate_bits = r;
  {
    const uint64_t wr_inv = gl::inv(gl::root(r)), sn_inv = gl::inv(gl::pow(gl::GENERATOR, N)), r_inv = gl::inv(R);
    for (uint32_t k = 0; k < R; k++) {
      ca.wr_inv_pow[k] = gl::pow(wr_inv, k);
      ca.chunk_scale[k] = gl::mulc(gl::pow(sn_inv, k), r_inv);
    }
  }
  TRY(launch_quotient_chunks(ca, st));
  (void)mark_q;
  // the temporaries are dead once the chunk kernel has run; stream order makes reuse safe
  w.arena.release(mark_tmp);
  Committed quot;
  TRY(commit(w, d_qc, Q, log_n, r, h, true, &quot));
  std::memcpy(P + L.quot_cap, quot.cap.data(), L.cap_words * 8);
  ch.observe(quot.cap.data(), L.cap_words);
  if (w.aborted()) return fail(BP_ERR_ABORTED, "aborted after quotient commitment");

  // 4. zeta
  const gl::Ext zeta = ch.challenge_ext();
  if (gl::eq(gl::pow(zeta, N), gl::ext(1))) return fail(BP_ERR_INVALID_INPUT, "Opening point is in the subgroup.");
  const gl::Ext zeta_next = gl::scale(zeta, wN);

  // 5. openings: dot products of bit-reversed coefficient columns with zeta^bitrev(pos)
  ARENA_ALLOC(d_pw, 6 * N);  // zeta, g zeta and 1 in one launch
  const uint64_t* d_pw1 = d_pw + 4 * N;
  TRY(launch_power_vectors(d_pw, log_n, zeta, zeta_next, 3, st, gl::ext(1)));
  const size_t open_cols = (size_t)K + C + A + Q + A;
  if (open_cols * 4 > w.pinned_words) return fail(BP_ERR_UNSUPPORTED, "too many columns for the opening mailbox");
  uint64_t* d_open = w.pinned_dev;  // kernels write the openings straight into host-visible memory
  {
    // one launch for the five opening sets (they used to be five launches in a row on the critical path)
    OpenMulti om{};
    om.stride = N; om.log_n = log_n;
    uint64_t* d_o = d_open;
    auto seg = [&](const uint64_t* coeffs, uint32_t n_cols, const uint64_t* pw, uint32_t n_points) {
      if (n_cols) {
        const uint32_t k = om.n_segs++;
        om.coeffs[k] = coeffs; om.pw[k] = pw; om.out[k] = d_o; om.n_points[k] = n_points;
        om.first_col[k + 1] = om.first_col[k] + n_cols;
      }
      d_o += (size_t)n_cols * 4;
    };
    seg(K ? consts->coeffs : nullptr, K, d_pw, 1);
    seg(trace.coeffs, C, d_pw, 2);
    seg(aux.coeffs, A, d_pw, 2);
    seg(quot.coeffs, Q, d_pw, 1);
    seg(aux.coeffs, A, d_pw1, 1);
    TRY(launch_openings_multi(om, st));
  }
  TRY(w.wait());
  std::vector<uint64_t> ho(w.pinned, w.pinned + open_cols * 4);
  {
    uint64_t* oz = P + L.open_zeta;
    for (size_t c = 0; c < (size_t)K + C + A + Q; c++) { oz[2 * c] = ho[4 * c]; oz[2 * c + 1] = ho[4 * c + 1]; }
    uint64_t* on = P + L.open_next;
    for (size_t c = 0; c < (size_t)C + A; c++) { on[2 * c] = ho[4 * (K + c) + 2]; on[2 * c + 1] = ho[4 * (K + c) + 3]; }
    uint64_t* of = P + L.open_first;
    const size_t base = (size_t)K + C + A + Q;
    for (size_t c = 0; c < A; c++) { of[2 * c] = ho[4 * (base + c)]; of[2 * c + 1] = ho[4 * (base + c) + 1]; }
  }
  ch.observe(P + L.open_zeta, 2 * (size_t)L.n_zeta);
  ch.observe(P + L.open_next, 2 * (size_t)L.n_next);
  ch.observe(P + L.open_first, 2 * (size_t)A);

  // 6. FRI.  Batches: (zeta: all), (g*zeta: trace, aux), (1: aux);  final = ((q0*a^k1 + q1)*a^k2 + q2)
  const gl::Ext alpha = ch.challenge_ext();
  const uint32_t k0 = L.n_zeta, k1 = L.n_next, k2 = A;
  ARENA_ALLOC(d_apow, 2 * (size_t)k0);
  TRY(launch_alpha_pows(d_apow, k0, alpha, st));
  struct OracleRef { const Committed* c; int32_t e[3]; };
  OracleRef refs[4];
  int n_or = 0;
  if (K) refs[n_or++] = {consts, {0, -1, -1}};
  refs[n_or++] = {&trace, {(int32_t)K, 0, -1}};
  refs[n_or++] = {&aux, {(int32_t)(K + C), (int32_t)C, 0}};
  refs[n_or++] = {&quot, {(int32_t)(K + C + A), -1, -1}};
  const uint32_t cols_per_chunk = 64;
  uint32_t total_chunks = 0;
  for (int o = 0; o < n_or; o++) total_chunks += (refs[o].c->n_cols + cols_per_chunk - 1) / cols_per_chunk;
  ARENA_ALLOC(d_cpart, (size_t)total_chunks * 6 * N);
  uint32_t chunk_base = 0;
  CombineMulti cm{};
  for (int o = 0; o < n_or; o++) {  // one launch over the chunks of all oracles
    CombineArgs& cb = cm.a[cm.n_oracles++];
    cb.coeffs = refs[o].c->coeffs; cb.stride = N; cb.log_n = log_n; cb.n_cols = refs[o].c->n_cols;
    cb.cols_per_chunk = cols_per_chunk; cb.chunk_base = chunk_base;
    for (int b = 0; b < 3; b++) cb.exp_base[b] = refs[o].e[b];
    cb.alpha_pows = d_apow; cb.partial = d_cpart;
    chunk_base += (cb.n_cols + cols_per_chunk - 1) / cols_per_chunk;
  }
  ARENA_ALLOC(d_g, 6 * N);
  if (device_loaded()) {  // several provers share the device: one pass, no partial sums, one launch instead of two
    TRY(launch_combine_all(cm, d_g, st));
  } else {
    TRY(launch_combine_partial_multi(cm, total_chunks, st));
    TRY(launch_combine_reduce(d_cpart, total_chunks, log_n, d_g, st));
  }
  ARENA_ALLOC(d_glde, 6 * M);
  TRY(ntt_br2nat(d_g, N, d_glde, M, N, log_n, 6, R, coset_scale, false, st));
  FriInitArgs fi{};
  fi.glde = d_glde; fi.tw_n = tw_n; fi.log_n = log_n; fi.rate_bits = r;
  std::memcpy(fi.g_t, g_t, sizeof(g_t));
  {
    const uint64_t* opens[3] = {P + L.open_zeta, P + L.open_next, P + L.open_first};
    const uint32_t kk[3] = {k0, k1, k2};
    for (int b = 0; b < 3; b++) {  // PrecomputedReducedOpenings: sum_j alpha^j opening_j
      gl::Ext acc = gl::ext(0);
      for (size_t j = kk[b]; j-- > 0;) acc = gl::add(gl::mul(acc, alpha), gl::Ext{opens[b][2 * j], opens[b][2 * j + 1]});
      fi.y[b] = acc;
    }
    fi.z[0] = zeta; fi.z[1] = zeta_next; fi.z[2] = gl::ext(1);
    fi.alpha_shift[0] = gl::pow(alpha, (uint64_t)k1 + k2);
    fi.alpha_shift[1] = gl::pow(alpha, k2);
    fi.alpha_shift[2] = gl::ext(1);
  }
  ARENA_ALLOC(d_v0, 2 * M);
  fi.out = d_v0;
  TRY(launch_fri_init(fi, st));

  // commit phase (fri_committed_trees), folding in the evaluation domain
  const uint32_t ab = cfg.arity_bits, arity = 1u << ab;
  const uint64_t *layer_values[9], *layer_digests[9];
  uint32_t layer_log_nl[9];
  uint64_t* cur = d_v0;
  uint32_t log_nl = log_n;
  uint64_t shift = gl::GENERATOR;
  for (uint32_t l = 0; l < L.n_layers; l++) {
    const uint32_t log_leaves = log_nl - ab + r;
    const size_t dw = bp_merkle_digest_words(log_leaves, h);
    ARENA_ALLOC(d_dig, dw);
    ARENA_ALLOC(d_next, (size_t)2 << log_leaves);
    FriLayerArgs fa{};
    TRY(fri_layer_args(log_nl, r, ab, shift, &fa));
    fa.values = cur; fa.out = d_next; fa.digests = d_dig;
    TRY(launch_fri_layer_leaves(fa, st));
    bool mirrored = false;
    TRY(merkle_upper_levels(d_dig, log_leaves, h, st, w.pinned_dev, &mirrored));
    uint64_t* cap = P + L.fri_caps + l * L.cap_words;
    if (mirrored) {
      TRY(w.wait());
      std::memcpy(cap, w.pinned, L.cap_words * 8);
    } else {
      TRY(w.d2h(cap, d_dig + dw - L.cap_words, L.cap_words));
    }
    ch.observe(cap, L.cap_words);
    fa.beta = ch.challenge_ext();
    TRY(launch_fri_fold(fa, st));
    layer_values[l] = cur; layer_digests[l] = d_dig; layer_log_nl[l] = log_nl;
    cur = d_next;
    log_nl -= ab;
    shift = gl::pow(shift, arity);
  }
  // final polynomial: interpolate the last layer on shift*<w> (tiny; host)
  {
    const uint32_t log_ml = log_nl + r;
    const uint64_t ml = (uint64_t)1 << log_ml, nl = (uint64_t)1 << log_nl;
    std::vector<uint64_t> hv(2 * ml);
    TRY(w.d2h(hv.data(), cur, hv.size()));
    const uint64_t w_inv = gl::inv(gl::root(log_ml)), s_inv = gl::inv(shift), ml_inv = gl::inv(ml);
    std::vector<uint64_t> wp(ml);  // w^-k
    wp[0] = 1;
    for (uint64_t k = 1; k < ml; k++) wp[k] = gl::mulc(wp[k - 1], w_inv);
    uint64_t sk = ml_inv;  // s^-k / m
    for (uint64_t k = 0; k < ml; k++) {
      uint64_t c0 = 0, c1 = 0;
      for (uint64_t i = 0; i < ml; i++) {  // natural index i = t + 2^r * m  <->  coset-major t*n_l + m
        const uint64_t pos = (i & (R - 1)) * nl + (i >> r);
        const uint64_t tw = wp[(i * k) & (ml - 1)];
        c0 = gl::addc(c0, gl::mulc(hv[2 * pos], tw));
        c1 = gl::addc(c1, gl::mulc(hv[2 * pos + 1], tw));
      }
      c0 = gl::mulc(c0, sk);
      c1 = gl::mulc(c1, sk);
      sk = gl::mulc(sk, s_inv);
      if (k < L.final_len) {
        P[L.final_poly + 2 * k] = c0;
        P[L.final_poly + 2 * k + 1] = c1;
      } else if (c0 || c1) {
        return fail(BP_ERR_INVALID_INPUT, "FRI final polynomial has a non-zero tail: the witness violates the AIR");
      }
    }
  }
  ch.observe(P + L.final_poly, 2 * (size_t)L.final_len);
  if (w.aborted()) return fail(BP_ERR_ABORTED, "aborted before proof of work");

  // proof of work: smallest witness
  {
    PowArgs pa{};
    ch.pow_state(pa.state, &pa.pos);
    pa.bits = cfg.pow_bits;
    uint64_t nonce = 0;
    if (cfg.pow_bits) {
      // The smallest witness is geometric with mean 2^pow_bits and every candidate costs a whole permutation, so
      // the batch size decides how many permutations are wasted past the winner: batches of 2^(pow_bits-1) stop
      // after 2.5 launches and 83 k permutations on average at 16 bits (one batch of 2 x 2^16 followed by
      // doubling: 168 k).  Proof of work is 7 % of a txn proof's VALU work (tools/valu_work_breakdown.py), the
      // extra launches are latency on one of 24 streams.  After 8 misses the batch grows (tiny bit counts, bad luck).
      const uint32_t batch0 = (uint32_t)std::min<uint64_t>(1u << 20, std::max<uint64_t>(1u << 12, (uint64_t)1 << (cfg.pow_bits ? cfg.pow_bits - 1 : 0)));
      uint32_t batch = batch0, n_batches = 0;
      unsigned long long res = ~0ULL;
      BPG_HIP(hipMemsetAsync(w.d_pow_result, 0xFF, 8, st));
      // Batches are launched four at a time before the host looks: a batch whose predecessors already found a
      // witness leaves at once (pow_grind_mx_kernel), so the speculation costs four tiny launches and saves the
      // device->host round trip after every batch (2.5 -> 1.2 round trips per proof at 16 bits).
      for (uint64_t base = 0;;) {
        for (int k = 0; k < 4; k++) {
          if (++n_batches > 8) batch = std::min<uint32_t>(1u << 20, batch * 2);
          pa.base = base;
          TRY(launch_pow(pa, batch, w.d_pow_result, st));
          base += batch;
        }
        TRY(w.d2h(reinterpret_cast<uint64_t*>(&res), reinterpret_cast<uint64_t*>(w.d_pow_result), 1));
        if (res != ~0ULL) break;
        if (w.aborted()) return fail(BP_ERR_ABORTED, "aborted during proof of work");
        if (base > ((uint64_t)1 << 44)) return fail(BP_ERR_DEVICE, "proof of work search exhausted");
      }
      nonce = res;
    }
    P[L.pow] = nonce;
    ch.observe(nonce);
    const uint64_t resp = ch.challenge();
    if (cfg.pow_bits && (resp >> (64 - cfg.pow_bits)) != 0)
      return fail(BP_ERR_DEVICE, "proof-of-work witness failed the host re-check");
  }

  // query phase: indices from the transcript, rows + Merkle paths gathered on the device
  {
    QueryArgs qa2{};
    QueryLayerArgs ql{};
    for (uint32_t q = 0; q < cfg.num_queries; q++) qa2.x_index[q] = ql.x_index[q] = ch.challenge() & (M - 1);
    const size_t q_words = (size_t)cfg.num_queries * L.query_words;
    const bool q_direct = q_words <= w.pinned_words;  // gather straight into host-visible memory when it fits
    uint64_t* d_q = q_direct ? w.pinned_dev : w.arena.alloc_words(q_words);
    if (!d_q) return fail(BP_ERR_DEVICE, "device arena exhausted (%zu MiB) allocating the query buffer", w.arena.capacity() >> 20);
    qa2.out = ql.out = d_q;
    qa2.query_words = ql.query_words = L.query_words;
    qa2.log_n = log_n; qa2.rate_bits = r; qa2.cap_height = h;
    uint32_t off = 1;
    for (int o = 0; o < n_or; o++) {
      const Committed* c = refs[o].c;
      qa2.oracle[o] = QueryOracle{c->lde, c->digests, M, c->n_cols, off};
      off += c->n_cols + L.depth0 * 4;
    }
    TRY(launch_query_initial(qa2, cfg.num_queries, n_or, st));
    ql.rate_bits = r; ql.cap_height = h; ql.arity_bits = ab;
    for (uint32_t l = 0; l < L.n_layers; l++) {
      ql.layer[l] = QueryLayer{layer_values[l], layer_digests[l], layer_log_nl[l], off};
      off += 2 * arity + (layer_log_nl[l] - ab + r - h) * 4;
    }
    TRY(launch_query_layers(ql, cfg.num_queries, L.n_layers, st));
    if (q_direct) {
      TRY(w.wait());
      std::memcpy(P + L.queries, w.pinned, q_words * 8);
    } else {
      TRY(w.d2h(P + L.queries, d_q, q_words));
    }
  }
  return BP_OK;
}

}  // namespace bpg
