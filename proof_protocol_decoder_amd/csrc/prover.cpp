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
#include <chrono>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <memory>
#include <thread>

namespace bpg {

// ------------------------------------------------------------------ host Poseidon
static const uint64_t RC_HOST[360] = {
#include "poseidon_rc.inc"
};
static inline uint64_t sbox_host(uint64_t x) {
  uint64_t x2 = gl::mulc(x, x), x4 = gl::mulc(x2, x2), x3 = gl::mulc(x2, x);
  return gl::mulc(x3, x4);
}
// The MDS layer over the 32-bit halves of the state words: the circulant's entries are below 2^6, so the twelve
// products of a row sum to less than 2^42 per half and the row is  lo_sum + 2^32 hi_sum  reduced once.  (The first
// version multiplied in 128 bits with a modulo in the index: 8..15 us per permutation, and a transaction's transcripts
// are ~5,800 sequential permutations on the proving thread.)  The sums vectorise four rows at a time where the CPU
// has AVX2 (vpmuludq takes the low halves as they lie in the 64-bit lanes); the scalar form is the fallback.
static const uint32_t MDS_C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static inline void mds_finish_host(const uint64_t al[12], const uint64_t ah[12], uint64_t s[12]) {
  for (int r = 0; r < 12; r++) {
    const uint64_t l = al[r] + (ah[r] << 32);
    const uint64_t h = (ah[r] >> 32) + (l < al[r]);
    s[r] = gl::canon(gl::reduce128(l, h));
  }
}
static inline void mds_host_scalar(uint64_t s[12]) {
  uint32_t lo[24], hi[24];
  for (int i = 0; i < 12; i++) {
    lo[i] = lo[i + 12] = (uint32_t)s[i];
    hi[i] = hi[i + 12] = (uint32_t)(s[i] >> 32);
  }
  uint64_t al[12], ah[12];
  for (int r = 0; r < 12; r++) {
    uint64_t a = 0, b = 0;
    for (int i = 0; i < 12; i++) {
      a += (uint64_t)lo[i + r] * MDS_C[i];
      b += (uint64_t)hi[i + r] * MDS_C[i];
    }
    al[r] = a;
    ah[r] = b;
  }
  al[0] += (uint64_t)lo[0] * 8;
  ah[0] += (uint64_t)hi[0] * 8;
  mds_finish_host(al, ah, s);
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) static inline void mds_host_avx2(uint64_t s[12]) {
  alignas(32) uint64_t w[24], h[24];
  for (int i = 0; i < 12; i++) {
    w[i] = w[i + 12] = s[i];
    h[i] = h[i + 12] = s[i] >> 32;
  }
  __m256i al[3], ah[3];
  for (int v = 0; v < 3; v++) al[v] = ah[v] = _mm256_setzero_si256();
  for (int i = 0; i < 12; i++) {
    const __m256i c = _mm256_set1_epi64x(MDS_C[i]);
    for (int v = 0; v < 3; v++) {
      al[v] = _mm256_add_epi64(al[v], _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(w + i + 4 * v)), c));
      ah[v] = _mm256_add_epi64(ah[v], _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(h + i + 4 * v)), c));
    }
  }
  alignas(32) uint64_t a[12], b[12];
  for (int v = 0; v < 3; v++) {
    _mm256_store_si256((__m256i*)(a + 4 * v), al[v]);
    _mm256_store_si256((__m256i*)(b + 4 * v), ah[v]);
  }
  a[0] += (uint64_t)(uint32_t)s[0] * 8;
  b[0] += (s[0] >> 32) * 8;
  mds_finish_host(a, b, s);
}
#endif
template <bool AVX2>
__attribute__((always_inline)) static inline void poseidon_host_body(uint64_t s[12]) {
  auto mds = [](uint64_t* st) {
#if defined(__x86_64__)
    if (AVX2) return mds_host_avx2(st);
#endif
    mds_host_scalar(st);
  };
  int rnd = 0;
  for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]);
  for (int phase = 0; phase < 3; phase++) {
    const int n_rounds = phase == 1 ? 22 : 4;
    for (int k = 0; k < n_rounds; k++, rnd++) {
      for (int i = 0; i < 12; i++) s[i] = gl::addc(s[i], RC_HOST[rnd * 12 + i]);
      if (phase == 1) s[0] = sbox_host(s[0]);
      else for (int i = 0; i < 12; i++) s[i] = sbox_host(s[i]);
      mds(s);
    }
  }
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void poseidon_host_avx2(uint64_t s[12]) { poseidon_host_body<true>(s); }
#endif
static void poseidon_host_scalar(uint64_t s[12]) { poseidon_host_body<false>(s); }
static std::atomic<int> g_host_poseidon{0};  // bp_tune_host_poseidon: 0 = by the CPU, 1 = scalar form (tests)
void tune_host_poseidon(int mode) { g_host_poseidon.store(mode == 1 ? 1 : 0); }
void poseidon_host(uint64_t s[12]) {
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2 && g_host_poseidon.load(std::memory_order_relaxed) == 0) return poseidon_host_avx2(s);
#endif
  poseidon_host_scalar(s);
}
// hash_no_pad of a recursion circuit's public-input list together with the WITNESS of its in-circuit computation (AIR
// 8's Poseidon gate, air.hpp): per absorbed chunk one row of air::plonk::H_WIRES wires -- the state in, the state out and
// the S-box inputs of every round in between.  rows: ceil(n / 8) x H_WIRES words; digest = the hash (the first four
// output words of the last row).
// one permutation with every S-box input kept: s in / out, w = the row's wires (H_FULL1, H_PART, H_FULL2; not in / out)
template <bool AVX2>
__attribute__((always_inline)) static inline void permutation_wires_body(uint64_t (&s)[12], uint64_t* w) {
  namespace pk = air::plonk;
  for (int rnd = 0; rnd < 30; rnd++) {
    for (int i = 0; i < 12; i++) s[i] = gl::addc(s[i], RC_HOST[rnd * 12 + i]);   // s = the S-box input of this round
    const bool full = rnd < 4 || rnd >= 26;
    if (rnd >= 1 && rnd <= 3) std::memcpy(w + pk::H_FULL1 + 12 * (rnd - 1), s, sizeof(s));
    else if (rnd >= 4 && rnd <= 25) w[pk::H_PART + rnd - 4] = s[0];
    else if (rnd >= 26) std::memcpy(w + pk::H_FULL2 + 12 * (rnd - 26), s, sizeof(s));
    if (full) for (int i = 0; i < 12; i++) s[i] = sbox_host(s[i]);
    else s[0] = sbox_host(s[0]);
#if defined(__x86_64__)
    if (AVX2) mds_host_avx2(s);
    else
#endif
      mds_host_scalar(s);
  }
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void permutation_wires_avx2(uint64_t (&s)[12], uint64_t* w) { permutation_wires_body<true>(s, w); }
#endif
static void permutation_wires(uint64_t (&s)[12], uint64_t* w) {
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2 && g_host_poseidon.load(std::memory_order_relaxed) == 0) return permutation_wires_avx2(s, w);
#endif
  permutation_wires_body<false>(s, w);
}
void poseidon_hash_rows(const uint64_t* in, size_t n, std::vector<uint64_t>* rows, uint64_t digest[4]) {
  namespace pk = air::plonk;
  const size_t H = (n + 7) / 8;
  rows->assign(H * pk::H_WIRES, 0);   // (swap and delta wires stay 0: a sponge row permutes what comes in)
  uint64_t s[12] = {0};
  for (size_t h = 0; h < H; h++) {
    uint64_t* w = rows->data() + h * pk::H_WIRES;
    const size_t k = std::min<size_t>(8, n - 8 * h);
    for (size_t i = 0; i < k; i++) s[i] = gl::canon(in[8 * h + i]);
    std::memcpy(w + pk::H_IN, s, sizeof(s));
    permutation_wires(s, w);
    std::memcpy(w + pk::H_OUT, s, sizeof(s));
  }
  std::memcpy(digest, s, 32);
}
// The witness of one Merkle path walked by Poseidon-gate rows (merkle_proofs::verify_merkle_proof_to_cap as a circuit):
// level l's row holds (node, sibling, 0) coming in, bit l of `index` on the swap wire, delta = bit (sibling - node), the
// permutation of (left, right, 0) and its output, whose first four words are the next level's node.  rows: depth x
// H_WIRES words; root = the last output (the cap entry the path arrives at).
void poseidon_merkle_rows(const uint64_t leaf[4], uint64_t index, const uint64_t* siblings, uint32_t depth, uint64_t* rows,
                          uint64_t root[4]) {
  namespace pk = air::plonk;
  uint64_t cur[4];
  for (int i = 0; i < 4; i++) cur[i] = gl::canon(leaf[i]);
  for (uint32_t l = 0; l < depth; l++, index >>= 1) {
    uint64_t* w = rows + (size_t)l * pk::H_WIRES;
    std::memset(w, 0, pk::H_WIRES * 8);
    const uint64_t* sib = siblings + 4 * (size_t)l;
    const uint64_t bit = index & 1;
    uint64_t s[12] = {0};
    for (int i = 0; i < 4; i++) {
      const uint64_t sv = gl::canon(sib[i]);
      w[pk::H_IN + i] = cur[i];
      w[pk::H_IN + 4 + i] = sv;
      const uint64_t d = bit ? gl::subc(sv, cur[i]) : 0;
      w[pk::H_DELTA + i] = d;
      s[i] = gl::addc(cur[i], d);
      s[4 + i] = gl::subc(sv, d);
    }
    w[pk::H_SWAP] = bit;
    permutation_wires(s, w);
    std::memcpy(w + pk::H_OUT, s, sizeof(s));
    std::memcpy(cur, s, 32);
  }
  std::memcpy(root, cur, 32);
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
  if (ai->n_cols && (c.n_cols != ai->n_cols || c.n_const != ai->n_const_max || c.deg_pow != (ai->degree > 3 ? 3u : 1u)))
    return fail(BP_ERR_INVALID_INPUT, "AIR %u (%s) has %u columns, %u constant columns and degree %u (deg_pow %u): got n_cols=%u "
                "n_const=%u deg_pow=%u", c.air_id, ai->name, ai->n_cols, ai->n_const_max, ai->degree, ai->degree > 3 ? 3u : 1u,
                c.n_cols, c.n_const, c.deg_pow);
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
  L.n_aux = air::ctl::n_aux(air::Shape{c.air_id, c.n_cols, c.n_const, c.deg_pow});
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

#define TRY(expr)        \
  do {                   \
    int _rc = (expr);    \
    if (_rc) return _rc; \
  } while (0)

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
  BPG_HIP(hipHostMalloc(reinterpret_cast<void**>(&hash_rows), MAX_BATCH * HASH_ROWS_WORDS * 8, hipHostMallocDefault));
  BPG_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&hash_rows_dev), hash_rows, 0));
  BPG_HIP(hipMalloc(reinterpret_cast<void**>(&d_pow_result), 8 * MAX_BATCH));
  // many prover threads share few host cores: wait on a blocking event (the thread sleeps) rather
  // than spin in hipStreamSynchronize
  BPG_HIP(hipEventCreateWithFlags(&sync_event, hipEventBlockingSync | hipEventDisableTiming));
  return init_ntt_kernels();
}
void Worker::destroy() {
  if (stream) (void)hipStreamSynchronize(stream);
  arena.destroy();
  if (pinned) (void)hipHostFree(pinned);
  if (hash_rows) (void)hipHostFree(hash_rows);
  hash_rows = hash_rows_dev = nullptr;
  if (d_pow_result) (void)hipFree(d_pow_result);
  if (sync_event) (void)hipEventDestroy(sync_event);
  if (stream) (void)hipStreamDestroy(stream);
  sync_event = nullptr;
  pinned = nullptr;
  d_pow_result = nullptr;
  stream = nullptr;
}
// Host waits.  Where the device has interrupt-driven waits (bp_host_wait_mode 1) the runtime's wait sleeps.  On a
// device the process had already used the library must not switch that mode (capi.cpp), and the runtime's wait spins
// at 100 % of a core -- twenty prover threads doing that starve the ones that have work (round 2: 18 instead of 25
// txn-proofs/s).  There the wait is the library's own: poll the event, and once the wait is older than a few
// microseconds sleep between polls, an eighth of the time waited so far (at most 200 us): the latency added to a
// stage is bounded by 1/8 of the stage, the CPU time of a waiting thread by the poll rate.
static std::atomic<int> g_host_wait{0};  // 0: by the device's mode; 1: always the runtime's wait; 2: always poll + sleep
void tune_host_wait(int mode) { g_host_wait.store(mode < 0 || mode > 2 ? 0 : mode); }
extern "C" int bp_host_wait_mode(int device);
int Worker::wait_recorded() {
  using clock = std::chrono::steady_clock;
  const int knob = g_host_wait.load(std::memory_order_relaxed);
  // While few provers are at work on the device (a lone transaction, the end of a shard, the aggregation tree's last
  // levels) every host wait is on the proof's critical path and there are cores to spare: poll without sleeping for
  // a while first -- an interrupt-driven wake-up costs 50..100 us, a poll sees the event within a few --, then wait
  // the usual way.  Under load (six or more provers) nobody spins: the cores belong to the threads that have work.
  if (knob != 1 && !device_loaded()) {
    const clock::time_point t0 = clock::now();
    for (;;) {
      const hipError_t e = hipEventQuery(sync_event);
      if (e == hipSuccess) return BP_OK;
      if (e != hipErrorNotReady) return fail(BP_ERR_DEVICE, "hipEventQuery failed: %s", hipGetErrorString(e));
      if (clock::now() - t0 > std::chrono::microseconds(1500)) break;
    }
  }
  if (knob == 1 || (knob == 0 && bp_host_wait_mode(device) == 1)) {
    BPG_HIP(hipEventSynchronize(sync_event));
    return BP_OK;
  }
  const clock::time_point t0 = clock::now();
  for (;;) {
    const hipError_t e = hipEventQuery(sync_event);
    if (e == hipSuccess) return BP_OK;
    if (e != hipErrorNotReady) return fail(BP_ERR_DEVICE, "hipEventQuery failed: %s", hipGetErrorString(e));
    const int64_t waited_us = std::chrono::duration_cast<std::chrono::microseconds>(clock::now() - t0).count();
    if (waited_us < 8) continue;  // the shortest stages (a copy, a tiny launch) finish in one or two polls
    std::this_thread::sleep_for(std::chrono::microseconds(std::min<int64_t>(200, std::max<int64_t>(5, waited_us / 8))));
  }
}
int Worker::wait() {
  BPG_HIP(hipEventRecord(sync_event, stream));
  return wait_recorded();
}
int Worker::d2h(uint64_t* host_dst, const uint64_t* dev_src, size_t words) {
  for (size_t off = 0; off < words; off += pinned_words) {
    size_t k = std::min(pinned_words, words - off);
    BPG_HIP(hipMemcpyAsync(pinned, dev_src + off, k * 8, hipMemcpyDeviceToHost, stream));
    TRY(wait());
    std::memcpy(host_dst + off, pinned, k * 8);
  }
  return BP_OK;
}

#define ARENA_ALLOC(var, words)                                                                         \
  uint64_t* var = w.arena.alloc_words(words);                                                           \
  if (!var)                                                                                             \
    return fail(BP_ERR_DEVICE, "device arena exhausted (%zu MiB) allocating %zu words for " #var,     \
                w.arena.capacity() >> 20, (size_t)(words))

extern "C" int bp_lde_batch(const uint64_t*, uint64_t, uint64_t*, uint64_t, uint64_t*, uint64_t, uint32_t, uint32_t,
                            uint32_t, int, void*);
extern "C" uint64_t bp_merkle_digest_words(uint32_t, uint32_t);

// `batch` commitments of one shape in lock-step: d_in holds the matrices one behind the other ([batch][n_cols][n],
// column stride n), the outputs are laid out the same way, and every launch -- inverse NTT, coset LDE, leaf hashing,
// Merkle levels -- covers all of them (the NTTs simply see batch * n_cols columns; the hash kernels take the tree
// index from grid.z).  out[b] are views into the shared buffers.  One host wait for all caps.
int commit_batch(Worker& w, const uint64_t* d_in, uint32_t n_cols, uint32_t batch, uint32_t log_n, uint32_t rate_bits,
                 uint32_t cap_height, bool from_coeffs, Committed* out) {
  PendingCommit pc;
  TRY(commit_launch(w, w, 0, d_in, n_cols, batch, log_n, rate_bits, cap_height, from_coeffs, &pc));
  return commit_finish(pc, out);
}
// The two halves of a commitment.  commit_launch allocates in `mem`'s arena and queues every launch on `lane`'s
// stream, the caps landing in lane's pinned mailbox from word `mailbox_word` on; commit_finish waits for the lane and
// reads them.  With mem == lane this is commit_batch; with another worker's stream as the lane, commitments that do not
// depend on each other (the seven trace commitments of a transaction) overlap on a device that is not loaded.
int commit_launch(Worker& mem, Worker& lane, size_t mailbox_word, const uint64_t* d_in, uint32_t n_cols, uint32_t batch,
                  uint32_t log_n, uint32_t rate_bits, uint32_t cap_height, bool from_coeffs, PendingCommit* pc) {
  Worker& w = mem;
  if (batch == 0 || batch > MAX_BATCH) return fail(BP_ERR_INVALID_INPUT, "commit: batch of %u", batch);
  const uint64_t n = (uint64_t)1 << log_n, m = n << rate_bits;
  const size_t all_cols = (size_t)n_cols * batch;
  ARENA_ALLOC(lde, all_cols * m);
  const size_t dw = bp_merkle_digest_words(log_n + rate_bits, cap_height);
  ARENA_ALLOC(digests, dw * batch);
  uint64_t* coeffs = const_cast<uint64_t*>(d_in);
  if (!from_coeffs) {
    ARENA_ALLOC(c, all_cols * n);
    coeffs = c;
  }
  TRY(bp_lde_batch(d_in, n, from_coeffs ? nullptr : coeffs, n, lde, m, log_n, rate_bits, (uint32_t)all_cols, from_coeffs,
                   lane.stream));
  // the kernel that makes the cap level writes it into the pinned mailbox as well: no copy launch
  const size_t cw = (size_t)4 << cap_height;
  if (mailbox_word + cw * batch > lane.pinned_words) return fail(BP_ERR_UNSUPPORTED, "caps do not fit the mailbox");
  bool mirrored = false;
  TRY(merkle_commit_cols(lde, m, n_cols, log_n, rate_bits, cap_height, digests, lane.stream, lane.pinned_dev + mailbox_word,
                         &mirrored, batch, (uint64_t)n_cols * m, dw));
  if (!mirrored) {  // the cap is the leaf level: no kernel above it that could have mirrored it
    for (uint32_t b = 0; b < batch; b++)
      BPG_HIP(hipMemcpyAsync(lane.pinned + mailbox_word + b * cw, digests + (b + 1) * dw - cw, cw * 8, hipMemcpyDeviceToHost,
                             lane.stream));
  }
  *pc = PendingCommit{&lane, mailbox_word, d_in, coeffs, lde, digests, dw, cw, n_cols, batch, log_n, rate_bits, cap_height, from_coeffs};
  return BP_OK;
}
int commit_finish(const PendingCommit& pc, Committed* out) {
  TRY(pc.lane->wait());
  const uint64_t n = (uint64_t)1 << pc.log_n, m = n << pc.rate_bits;
  for (uint32_t b = 0; b < pc.batch; b++) {
    Committed& o = out[b];
    o.log_n = pc.log_n; o.n_cols = pc.n_cols; o.rate_bits = pc.rate_bits; o.cap_height = pc.cap_height;
    o.coeffs = pc.coeffs + (size_t)b * pc.n_cols * n;
    o.lde = pc.lde + (size_t)b * pc.n_cols * m;
    o.digests = pc.digests + (size_t)b * pc.dw;
    o.values = pc.from_coeffs ? nullptr : pc.d_in + (size_t)b * pc.n_cols * n;
    const uint64_t* cap = pc.lane->pinned + pc.mailbox_word + b * pc.cw;
    o.cap.assign(cap, cap + pc.cw);
  }
  return BP_OK;
}
int commit(Worker& w, const uint64_t* d_in, uint32_t n_cols, uint32_t log_n, uint32_t rate_bits,
           uint32_t cap_height, bool from_coeffs, Committed* out) {
  return commit_batch(w, d_in, n_cols, 1, log_n, rate_bits, cap_height, from_coeffs, out);
}

// Everything of the quotient launch that follows from the table shape and the challenges (the caller
// sets the three LDE pointers and the two output buffers).  Used by stark_prove and bp_quotient_eval.
int quotient_args(const StarkCfg& cfg, const Ctl& ctl, uint64_t alpha0, uint64_t alpha1, QuotArgs* out, QuotCoset* coset,
                  int loaded, uint32_t batch) {
  QuotArgs& qa = *out;
  const uint32_t log_n = cfg.log_n, r = cfg.rate_bits, R = 1u << r, C = cfg.n_cols, K = cfg.n_const;
  const air::Shape shape{cfg.air_id, C, K, cfg.deg_pow};
  const uint32_t A = air::ctl::n_aux(shape);
  const uint64_t N = (uint64_t)1 << log_n, M = N << r;
  const uint64_t wM = gl::root(log_n + r), wN = gl::root(log_n);
  qa.trace_stride = qa.aux_stride = qa.const_stride = M;
  TRY(get_table(0, log_n, 0, &qa.tw_n));
  qa.air_id = cfg.air_id;
  qa.log_n = log_n; qa.rate_bits = r; qa.n_cols = C; qa.n_const = K; qa.n_aux = A; qa.deg_pow = cfg.deg_pow;
  // the constraint list and its units (air.hpp): the AIR's, then the table's lookups (air::ctl).  The synthetic
  // table's many product columns are sliced into units; a real table's few lookup columns are one unit.
  qa.n_air_constraints = air::n_constraints(shape);
  qa.n_constraints = qa.n_air_constraints + air::ctl::n_constraints(shape);
  qa.side_rows = cfg.air_id == air::PLONK ? 1 : 0;
  qa.n_air_units = air::n_units(shape) - qa.side_rows;
  qa.aux_per_unit = cfg.air_id == air::SYNTHETIC ? std::max<uint32_t>(16, (A + 15) / 16) : A;
  // (AIR 8's copy constraints ride in its own ten units, next to the gates that read the same wires: no lookup unit)
  qa.n_ctl_units = cfg.air_id == air::PLONK ? 0 : (A + qa.aux_per_unit - 1) / qa.aux_per_unit;
  // One pass (the alpha fold never leaves the registers) once the rows alone fill the chip: 2048 workgroups of
  // 256 lanes = 2 per SIMD.  Shorter tables spread their units over grid.y until the launch has that many.
  // While several provers share the device nothing needs filling: one pass, one launch fewer on the proof's
  // critical path and no partial sums -- for the synthetic AIR, whose row costs little.  The AIRs of the real tables
  // cost 10^4 .. 10^5 instructions per row: one pass over a SHORT table is then milliseconds of a few workgroups on the
  // critical path (Keccak sponge, 2^9 rows: 2.6 ms against 0.11 ms spread, profiles/r3_k5_air_probe.txt), so their
  // units are spread until the launch has 256 workgroups.  The proofs of a batch (grid.z) count as rows of the launch.
  const uint32_t n_units = qa.n_air_units + qa.n_ctl_units, wg_x = (uint32_t)((M + 255) / 256) * std::max<uint32_t>(1, batch);
  const uint32_t loaded_rows = (cfg.air_id == air::SYNTHETIC && !g_k5_spread_all.load(std::memory_order_relaxed))
                                   ? 1 : std::min<uint32_t>(n_units, std::max<uint32_t>(1, (256 + wg_x - 1) / wg_x));
  const bool is_loaded = loaded < 0 ? device_loaded() : loaded != 0;
  const uint32_t want_rows = is_loaded ? loaded_rows : std::min<uint32_t>(n_units, std::max<uint32_t>(1, (2048 + wg_x - 1) / wg_x));
  qa.units_per_wg = (n_units + want_rows - 1) / want_rows;
  qa.alpha0 = alpha0; qa.alpha1 = alpha1;
  qa.g = wN; qa.g_inv = gl::inv(wN); qa.n_inv = gl::inv(N);
  // per-coset constants: g_t = 7 * w_M^t, Z_H(g_t x) = g_t^n - 1 (constant on a coset)
  if (coset)
    for (uint32_t t = 0; t < R; t++) {
      coset->g_t[t] = gl::mulc(gl::GENERATOR, gl::pow(wM, t));
      coset->zh_t[t] = gl::subc(gl::pow(coset->g_t[t], N), 1);
      coset->zh_inv_t[t] = gl::inv(coset->zh_t[t]);
    }
  qa.ctl = ctl;
  return BP_OK;
}
size_t quotient_partial_words(const QuotArgs& qa) {
  const uint32_t n_units = qa.n_air_units + qa.n_ctl_units, wg_rows = (n_units + qa.units_per_wg - 1) / qa.units_per_wg + qa.side_rows;
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

// ------------------------------------------------------------------ one table, or a batch of equally shaped ones
int stark_prove(Worker& w, const StarkCfg& cfg, const Committed* consts, const Committed& trace,
                const uint64_t* d_tv, const Ctl& ctl, Challenger& ch, std::vector<uint64_t>& proof, uint64_t* first_trace_leaf) {
  return stark_prove_batch(w, cfg, 1, &consts, &trace, &d_tv, &ctl, &ch, &proof, first_trace_leaf);
}

int stark_prove_batch(Worker& w, const StarkCfg& cfg, uint32_t B, const Committed* const* consts,
                      const Committed* trace, const uint64_t* const* d_tv, const Ctl* ctl, Challenger* ch,
                      std::vector<uint64_t>* proofs, uint64_t* first_trace_leaf) {
  TRY(check_cfg(cfg));
  if (B == 0 || B > MAX_BATCH || (size_t)B * cfg.num_queries > MAX_BATCH_QUERIES)
    return fail(BP_ERR_INVALID_INPUT, "stark_prove_batch: %u proofs x %u queries (at most %u proofs, %u queries in all)", B,
                cfg.num_queries, MAX_BATCH, MAX_BATCH_QUERIES);
  const ProofLayout L = proof_layout(cfg);
  const uint32_t log_n = cfg.log_n, r = cfg.rate_bits, h = cfg.cap_height, R = 1u << r;
  const uint64_t N = (uint64_t)1 << log_n, M = N << r;
  const uint32_t C = cfg.n_cols, K = cfg.n_const, A = L.n_aux, Q = L.n_quot;
  hipStream_t st = w.stream;
  uint64_t* P[MAX_BATCH];
  for (uint32_t b = 0; b < B; b++) {
    if (K && !(consts && consts[b])) return fail(BP_ERR_INVALID_INPUT, "constants commitment missing");
    proofs[b].assign(L.total, 0);
    uint64_t* p = P[b] = proofs[b].data();
    p[0] = PROOF_MAGIC; p[1] = log_n; p[2] = C; p[3] = K; p[4] = A; p[5] = Q; p[6] = r; p[7] = h;
    p[8] = cfg.num_queries; p[9] = L.n_layers; p[10] = L.final_len; p[11] = cfg.deg_pow; p[12] = cfg.pow_bits;
    p[13] = cfg.arity_bits; p[14] = cfg.air_id;
    std::memcpy(p + L.trace_cap, trace[b].cap.data(), L.cap_words * 8);
  }
  if (w.aborted()) return fail(BP_ERR_ABORTED, "aborted before auxiliary commitment");

  // per-coset constants: g_t = 7 * w_M^t, Z_H(g_t x) = g_t^n - 1 (constant on a coset)
  const uint64_t wN = gl::root(log_n);
  const uint64_t *tw_n = nullptr, *coset_scale = nullptr, *coset_scale_inv = nullptr;
  TRY(get_table(0, log_n, 0, &tw_n));
  TRY(get_table(2, log_n, r, &coset_scale));
  TRY(get_table(3, log_n, r, &coset_scale_inv));

  // 1. auxiliary columns -- the table's lookups (air::ctl): helper columns and running products -- and their commitment
  ARENA_ALLOC(d_auxv, (size_t)B * A * N);
  {
    AuxArgs aa[MAX_BATCH];
    for (uint32_t b = 0; b < B; b++) {
      aa[b] = AuxArgs{d_tv[b], d_auxv + (size_t)b * A * N, ctl[b]};
      if (K) aa[b].consts = consts[b]->values;
    }
    TRY(launch_aux(aa, B, cfg.air_id, C, log_n, st));
  }
  Committed aux[MAX_BATCH];
  TRY(commit_batch(w, d_auxv, A, B, log_n, r, h, false, aux));
  for (uint32_t b = 0; b < B; b++) {
    std::memcpy(P[b] + L.aux_cap, aux[b].cap.data(), L.cap_words * 8);
    ch[b].observe(aux[b].cap.data(), L.cap_words);
  }

  // 2. alphas;  3. quotient on the LDE coset -> per-coset iNTT -> chunk coefficients -> commitment
  QuotArgs qa[MAX_BATCH] = {};
  QuotCoset coset{};
  const int loaded = device_loaded();  // one answer for the whole batch: the proofs share one grid
  for (uint32_t b = 0; b < B; b++) {
    const uint64_t alpha0 = ch[b].challenge(), alpha1 = ch[b].challenge();
    qa[b].trace_lde = trace[b].lde; qa[b].aux_lde = aux[b].lde; qa[b].const_lde = K ? consts[b]->lde : nullptr;
    TRY(quotient_args(cfg, ctl[b], alpha0, alpha1, &qa[b], &coset, loaded, B));
  }
  ARENA_ALLOC(d_qc, (size_t)B * Q * N);  // chunk coefficients: live until the end (quotient oracle)
  const size_t mark_tmp = w.arena.mark();
  {
    const size_t cpow_words = 2 * (size_t)qa[0].n_constraints + 48, part_words = quotient_partial_words(qa[0]) + 1;
    ARENA_ALLOC(d_cpow, B * cpow_words);
    ARENA_ALLOC(d_partial, B * part_words);
    ARENA_ALLOC(d_qvals, (size_t)B * 2 * M);
    for (uint32_t b = 0; b < B; b++) {
      qa[b].apow = d_cpow + b * cpow_words; qa[b].partial = d_partial + b * part_words; qa[b].qvals = d_qvals + (size_t)b * 2 * M;
    }
    TRY(launch_quotient(qa, B, coset, st));
    TRY(intt_nat2br(d_qvals, N, d_qvals, N, log_n, B * 2 * R, true, st));  // 2 challenges x 2^r cosets per proof, in place
    ChunkArgs ca[MAX_BATCH] = {};
    const uint64_t wr_inv = gl::inv(gl::root(r)), sn_inv = gl::inv(gl::pow(gl::GENERATOR, N)), r_inv = gl::inv(R);
    for (uint32_t b = 0; b < B; b++) {
      ChunkArgs& c = ca[b];
      c.e = d_qvals + (size_t)b * 2 * M; c.inv_scale = coset_scale_inv; c.out = d_qc + (size_t)b * Q * N; c.out_stride = N;
      c.log_n = log_n; c.rate_bits = r;
      for (uint32_t k = 0; k < R; k++) {
        c.wr_inv_pow[k] = gl::pow(wr_inv, k);
        c.chunk_scale[k] = gl::mulc(gl::pow(sn_inv, k), r_inv);
      }
    }
    TRY(launch_quotient_chunks(ca, B, st));
  }
  // the temporaries are dead once the chunk kernel has run; stream order makes reuse safe
  w.arena.release(mark_tmp);
  Committed quot[MAX_BATCH];
  TRY(commit_batch(w, d_qc, Q, B, log_n, r, h, true, quot));
  for (uint32_t b = 0; b < B; b++) {
    std::memcpy(P[b] + L.quot_cap, quot[b].cap.data(), L.cap_words * 8);
    ch[b].observe(quot[b].cap.data(), L.cap_words);
  }
  if (w.aborted()) return fail(BP_ERR_ABORTED, "aborted after quotient commitment");

  // 4. zeta
  gl::Ext zeta[MAX_BATCH], zeta_next[MAX_BATCH];
  for (uint32_t b = 0; b < B; b++) {
    zeta[b] = ch[b].challenge_ext();
    if (gl::eq(gl::pow(zeta[b], N), gl::ext(1))) return fail(BP_ERR_INVALID_INPUT, "Opening point is in the subgroup.");
    zeta_next[b] = gl::scale(zeta[b], wN);
  }

  // 5. openings: dot products of bit-reversed coefficient columns with zeta^bitrev(pos)
  ARENA_ALLOC(d_pw, (size_t)B * 6 * N);  // zeta, g zeta and 1 of every proof in one launch
  {
    PowerVecArgs pv[MAX_BATCH];
    for (uint32_t b = 0; b < B; b++) pv[b] = PowerVecArgs{d_pw + (size_t)b * 6 * N, {zeta[b], zeta_next[b], gl::ext(1)}};
    TRY(launch_power_vectors(pv, B, log_n, 3, st));
  }
  const size_t open_cols = (size_t)K + C + A + Q + A;
  if (open_cols * 4 * B > w.pinned_words) return fail(BP_ERR_UNSUPPORTED, "too many columns for the opening mailbox");
  {
    // one launch for the five opening sets of every proof (they used to be five launches in a row on the critical
    // path); the kernels write the openings straight into host-visible memory
    OpenMulti om[MAX_BATCH] = {};
    for (uint32_t b = 0; b < B; b++) {
      OpenMulti& o = om[b];
      o.stride = N; o.log_n = log_n;
      const uint64_t *pw = d_pw + (size_t)b * 6 * N, *pw1 = pw + 4 * N;
      uint64_t* d_o = w.pinned_dev + (size_t)b * open_cols * 4;
      auto seg = [&](const uint64_t* coeffs, uint32_t n_cols, const uint64_t* pwv, uint32_t n_points) {
        if (n_cols) {
          const uint32_t k = o.n_segs++;
          o.coeffs[k] = coeffs; o.pw[k] = pwv; o.out[k] = d_o; o.n_points[k] = n_points;
          o.first_col[k + 1] = o.first_col[k] + n_cols;
        }
        d_o += (size_t)n_cols * 4;
      };
      seg(K ? consts[b]->coeffs : nullptr, K, pw, 1);
      seg(trace[b].coeffs, C, pw, 2);
      seg(aux[b].coeffs, A, pw, 2);
      seg(quot[b].coeffs, Q, pw, 1);
      seg(aux[b].coeffs, A, pw1, 1);
    }
    TRY(launch_openings_multi(om, B, st));
  }
  TRY(w.wait());
  for (uint32_t b = 0; b < B; b++) {
    const uint64_t* ho = w.pinned + (size_t)b * open_cols * 4;
    uint64_t* oz = P[b] + L.open_zeta;
    for (size_t c = 0; c < (size_t)K + C + A + Q; c++) { oz[2 * c] = ho[4 * c]; oz[2 * c + 1] = ho[4 * c + 1]; }
    uint64_t* on = P[b] + L.open_next;
    for (size_t c = 0; c < (size_t)C + A; c++) { on[2 * c] = ho[4 * (K + c) + 2]; on[2 * c + 1] = ho[4 * (K + c) + 3]; }
    uint64_t* of = P[b] + L.open_first;
    const size_t base = (size_t)K + C + A + Q;
    for (size_t c = 0; c < A; c++) { of[2 * c] = ho[4 * (base + c)]; of[2 * c + 1] = ho[4 * (base + c) + 1]; }
    ch[b].observe(P[b] + L.open_zeta, 2 * (size_t)L.n_zeta);
    ch[b].observe(P[b] + L.open_next, 2 * (size_t)L.n_next);
    ch[b].observe(P[b] + L.open_first, 2 * (size_t)A);
  }

  // 6. FRI.  Batches: (zeta: all), (g*zeta: trace, aux), (1: aux);  final = ((q0*a^k1 + q1)*a^k2 + q2)
  gl::Ext alpha[MAX_BATCH];
  for (uint32_t b = 0; b < B; b++) alpha[b] = ch[b].challenge_ext();
  const uint32_t k0 = L.n_zeta, k1 = L.n_next, k2 = A;
  ARENA_ALLOC(d_apow, (size_t)B * 2 * k0);
  {
    AlphaPowArgs ap[MAX_BATCH];
    for (uint32_t b = 0; b < B; b++) ap[b] = AlphaPowArgs{d_apow + (size_t)b * 2 * k0, alpha[b]};
    TRY(launch_alpha_pows(ap, B, k0, st));
  }
  struct OracleRef { const Committed* c; int32_t e[3]; };
  OracleRef refs[MAX_BATCH][4];
  int n_or = 0;
  for (uint32_t b = 0; b < B; b++) {
    n_or = 0;
    if (K) refs[b][n_or++] = {consts[b], {0, -1, -1}};
    refs[b][n_or++] = {&trace[b], {(int32_t)K, 0, -1}};
    refs[b][n_or++] = {&aux[b], {(int32_t)(K + C), (int32_t)C, 0}};
    refs[b][n_or++] = {&quot[b], {(int32_t)(K + C + A), -1, -1}};
  }
  const uint32_t cols_per_chunk = 64;
  uint32_t total_chunks = 0;
  for (int o = 0; o < n_or; o++) total_chunks += (refs[0][o].c->n_cols + cols_per_chunk - 1) / cols_per_chunk;
  ARENA_ALLOC(d_g, (size_t)B * 6 * N);
  {
    const bool one_pass = loaded != 0;  // several provers share the device: one pass, no partial sums, one launch instead of two
    uint64_t* d_cpart = nullptr;
    if (!one_pass) {
      ARENA_ALLOC(cp, (size_t)B * total_chunks * 6 * N);
      d_cpart = cp;
    }
    CombineMulti cm[MAX_BATCH] = {};
    uint64_t* gp[MAX_BATCH];
    CombineReduceArgs cr[MAX_BATCH];
    for (uint32_t b = 0; b < B; b++) {
      uint32_t chunk_base = 0;
      uint64_t* part = d_cpart ? d_cpart + (size_t)b * total_chunks * 6 * N : nullptr;
      for (int o = 0; o < n_or; o++) {  // one launch over the chunks of all oracles
        CombineArgs& cb = cm[b].a[cm[b].n_oracles++];
        cb.coeffs = refs[b][o].c->coeffs; cb.stride = N; cb.log_n = log_n; cb.n_cols = refs[b][o].c->n_cols;
        cb.cols_per_chunk = cols_per_chunk; cb.chunk_base = chunk_base;
        for (int k = 0; k < 3; k++) cb.exp_base[k] = refs[b][o].e[k];
        cb.alpha_pows = d_apow + (size_t)b * 2 * k0; cb.partial = part;
        chunk_base += (cb.n_cols + cols_per_chunk - 1) / cols_per_chunk;
      }
      gp[b] = d_g + (size_t)b * 6 * N;
      cr[b] = CombineReduceArgs{part, gp[b]};
    }
    if (one_pass) {
      TRY(launch_combine_all(cm, B, gp, st));
    } else {
      TRY(launch_combine_partial_multi(cm, B, total_chunks, st));
      TRY(launch_combine_reduce(cr, B, total_chunks, log_n, st));
    }
  }
  ARENA_ALLOC(d_glde, (size_t)B * 6 * M);
  TRY(ntt_br2nat(d_g, N, d_glde, M, N, log_n, 6 * B, R, coset_scale, false, st));
  ARENA_ALLOC(d_v0, (size_t)B * 2 * M);
  {
    FriInitArgs fi[MAX_BATCH] = {};
    for (uint32_t b = 0; b < B; b++) {
      FriInitArgs& f = fi[b];
      f.glde = d_glde + (size_t)b * 6 * M; f.tw_n = tw_n; f.log_n = log_n; f.rate_bits = r;
      std::memcpy(f.g_t, coset.g_t, sizeof(f.g_t));
      const uint64_t* opens[3] = {P[b] + L.open_zeta, P[b] + L.open_next, P[b] + L.open_first};
      const uint32_t kk[3] = {k0, k1, k2};
      for (int k = 0; k < 3; k++) {  // PrecomputedReducedOpenings: sum_j alpha^j opening_j
        gl::Ext acc = gl::ext(0);
        for (size_t j = kk[k]; j-- > 0;) acc = gl::add(gl::mul(acc, alpha[b]), gl::Ext{opens[k][2 * j], opens[k][2 * j + 1]});
        f.y[k] = acc;
      }
      f.z[0] = zeta[b]; f.z[1] = zeta_next[b]; f.z[2] = gl::ext(1);
      f.alpha_shift[0] = gl::pow(alpha[b], (uint64_t)k1 + k2);
      f.alpha_shift[1] = gl::pow(alpha[b], k2);
      f.alpha_shift[2] = gl::ext(1);
      f.out = d_v0 + (size_t)b * 2 * M;
    }
    TRY(launch_fri_init(fi, B, st));
  }

  // commit phase (fri_committed_trees), folding in the evaluation domain
  const uint32_t ab = cfg.arity_bits, arity = 1u << ab;
  // per layer: the values and digests of proof b sit b * (stride) words behind those of proof 0
  const uint64_t *layer_values[9], *layer_digests[9];
  size_t layer_vstride[9], layer_dstride[9];
  uint32_t layer_log_nl[9];
  uint64_t* cur = d_v0;
  size_t cur_stride = 2 * M;
  uint32_t log_nl = log_n;
  uint64_t shift = gl::GENERATOR;
  for (uint32_t l = 0; l < L.n_layers; l++) {
    const uint32_t log_leaves = log_nl - ab + r;
    const size_t dw = bp_merkle_digest_words(log_leaves, h), next_words = (size_t)2 << log_leaves;
    ARENA_ALLOC(d_dig, dw * B);
    ARENA_ALLOC(d_next, next_words * B);
    FriLayerArgs fa[MAX_BATCH] = {};
    TRY(fri_layer_args(log_nl, r, ab, shift, &fa[0]));
    for (uint32_t b = 0; b < B; b++) {
      if (b) fa[b] = fa[0];
      fa[b].values = cur + b * cur_stride; fa[b].out = d_next + b * next_words; fa[b].digests = d_dig + b * dw;
    }
    TRY(launch_fri_layer_leaves(fa, B, st));
    bool mirrored = false;
    TRY(merkle_upper_levels(d_dig, log_leaves, h, st, w.pinned_dev, &mirrored, B, dw));
    if (!mirrored)
      for (uint32_t b = 0; b < B; b++)
        BPG_HIP(hipMemcpyAsync(w.pinned + b * L.cap_words, d_dig + (b + 1) * dw - L.cap_words, L.cap_words * 8,
                               hipMemcpyDeviceToHost, st));
    TRY(w.wait());
    for (uint32_t b = 0; b < B; b++) {
      uint64_t* cap = P[b] + L.fri_caps + l * L.cap_words;
      std::memcpy(cap, w.pinned + b * L.cap_words, L.cap_words * 8);
      ch[b].observe(cap, L.cap_words);
      fa[b].beta = ch[b].challenge_ext();
    }
    TRY(launch_fri_fold(fa, B, st));
    layer_values[l] = cur; layer_vstride[l] = cur_stride; layer_digests[l] = d_dig; layer_dstride[l] = dw;
    layer_log_nl[l] = log_nl;
    cur = d_next;
    cur_stride = next_words;
    log_nl -= ab;
    shift = gl::pow(shift, arity);
  }
  // final polynomial: interpolate the last layer on shift*<w> (tiny; host)
  {
    const uint32_t log_ml = log_nl + r;
    const uint64_t ml = (uint64_t)1 << log_ml, nl = (uint64_t)1 << log_nl;
    if (2 * ml * B > w.pinned_words) return fail(BP_ERR_UNSUPPORTED, "final FRI layer does not fit the mailbox");
    for (uint32_t b = 0; b < B; b++)
      BPG_HIP(hipMemcpyAsync(w.pinned + b * 2 * ml, cur + b * cur_stride, 2 * ml * 8, hipMemcpyDeviceToHost, st));
    TRY(w.wait());
    const uint64_t w_inv = gl::inv(gl::root(log_ml)), s_inv = gl::inv(shift), ml_inv = gl::inv(ml);
    std::vector<uint64_t> wp(ml);  // w^-k
    wp[0] = 1;
    for (uint64_t k = 1; k < ml; k++) wp[k] = gl::mulc(wp[k - 1], w_inv);
    for (uint32_t b = 0; b < B; b++) {
      const uint64_t* hv = w.pinned + b * 2 * ml;
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
          P[b][L.final_poly + 2 * k] = c0;
          P[b][L.final_poly + 2 * k + 1] = c1;
        } else if (c0 || c1) {
          return fail(BP_ERR_INVALID_INPUT, "FRI final polynomial has a non-zero tail: the witness violates the AIR");
        }
      }
      ch[b].observe(P[b] + L.final_poly, 2 * (size_t)L.final_len);
    }
  }
  if (w.aborted()) return fail(BP_ERR_ABORTED, "aborted before proof of work");

  // proof of work: smallest witness
  {
    PowArgs pa[MAX_BATCH] = {};
    for (uint32_t b = 0; b < B; b++) {
      ch[b].pow_state(pa[b].state, &pa[b].pos);
      pa[b].bits = cfg.pow_bits;
    }
    unsigned long long res[MAX_BATCH];
    for (uint32_t b = 0; b < B; b++) res[b] = 0;
    if (cfg.pow_bits) {
      // The smallest witness is geometric with mean 2^pow_bits and every candidate costs a whole permutation, so
      // the batch size decides how many permutations are wasted past the winner: batches of 2^(pow_bits-1) stop
      // after 2.5 launches and 83 k permutations on average at 16 bits (one batch of 2 x 2^16 followed by
      // doubling: 168 k).  Proof of work is 3.5 % of a txn proof's VALU work (profiles/r3_sq_loaded_by_kernel.txt), the
      // extra launches are latency on one of 24 streams.  After 8 misses the batch grows (tiny bit counts, bad luck).
      const uint32_t batch0 = (uint32_t)std::min<uint64_t>(1u << 20, std::max<uint64_t>(1u << 12, (uint64_t)1 << (cfg.pow_bits ? cfg.pow_bits - 1 : 0)));
      uint32_t batch = batch0, n_batches = 0;
      BPG_HIP(hipMemsetAsync(w.d_pow_result, 0xFF, 8 * MAX_BATCH, st));
      // Batches are launched four at a time before the host looks: a batch whose predecessors already found a
      // witness leaves at once (pow_grind_mx_kernel), so the speculation costs four tiny launches and saves the
      // device->host round trip after every batch (2.5 -> 1.2 round trips per proof at 16 bits).  The proofs of a
      // batch search side by side (grid.z), each for its own witness: a proof that has found its own leaves at once.
      for (uint64_t base = 0;;) {
        for (int k = 0; k < 4; k++) {
          if (++n_batches > 8) batch = std::min<uint32_t>(1u << 20, batch * 2);
          for (uint32_t b = 0; b < B; b++) pa[b].base = base;
          TRY(launch_pow(pa, B, batch, w.d_pow_result, st));
          base += batch;
        }
        BPG_HIP(hipMemcpyAsync(w.pinned, w.d_pow_result, 8 * B, hipMemcpyDeviceToHost, st));
        TRY(w.wait());
        bool all = true;
        for (uint32_t b = 0; b < B; b++) {
          res[b] = reinterpret_cast<const unsigned long long*>(w.pinned)[b];
          all = all && res[b] != ~0ULL;
        }
        if (all) break;
        if (w.aborted()) return fail(BP_ERR_ABORTED, "aborted during proof of work");
        if (base > ((uint64_t)1 << 44)) return fail(BP_ERR_DEVICE, "proof of work search exhausted");
      }
    }
    for (uint32_t b = 0; b < B; b++) {
      const uint64_t nonce = res[b];
      P[b][L.pow] = nonce;
      ch[b].observe(nonce);
      const uint64_t resp = ch[b].challenge();
      if (cfg.pow_bits && (resp >> (64 - cfg.pow_bits)) != 0)
        return fail(BP_ERR_DEVICE, "proof-of-work witness failed the host re-check");
    }
  }

  // query phase: indices from the transcript, rows + Merkle paths gathered on the device
  {
    // (heap: the two argument blocks are 3 KiB each)
    std::unique_ptr<QueryArgs> qa2(new QueryArgs());
    std::unique_ptr<QueryLayerArgs> ql(new QueryLayerArgs());
    const size_t q_words = (size_t)cfg.num_queries * L.query_words;
    const bool q_direct = q_words * B + 4 * B <= w.pinned_words;  // gather straight into host-visible memory when it fits
    uint64_t* d_q = q_direct ? w.pinned_dev : w.arena.alloc_words(q_words * B + 4 * B);  // (+ the first leaf digests, at the end)
    if (!d_q) return fail(BP_ERR_DEVICE, "device arena exhausted (%zu MiB) allocating the query buffer", w.arena.capacity() >> 20);
    qa2->query_words = ql->query_words = L.query_words;
    qa2->n_queries = ql->n_queries = cfg.num_queries;
    qa2->log_n = log_n; qa2->rate_bits = r; qa2->cap_height = h;
    ql->rate_bits = r; ql->cap_height = h; ql->arity_bits = ab;
    for (uint32_t b = 0; b < B; b++) {
      for (uint32_t q = 0; q < cfg.num_queries; q++)
        qa2->x_index[b * cfg.num_queries + q] = ql->x_index[b * cfg.num_queries + q] = ch[b].challenge() & (M - 1);
      qa2->proof[b].out = ql->proof[b].out = d_q + b * q_words;
      qa2->proof[b].first_leaf = first_trace_leaf ? d_q + B * q_words + 4 * b : nullptr;
      uint32_t off = 1;
      for (int o = 0; o < n_or; o++) {
        const Committed* c = refs[b][o].c;
        qa2->proof[b].oracle[o] = QueryOracle{c->lde, c->digests, M, c->n_cols, off};
        off += c->n_cols + L.depth0 * 4;
      }
      for (uint32_t l = 0; l < L.n_layers; l++) {
        ql->proof[b].layer[l] = QueryLayer{layer_values[l] + b * layer_vstride[l], layer_digests[l] + b * layer_dstride[l],
                                          layer_log_nl[l], off};
        off += 2 * arity + (layer_log_nl[l] - ab + r - h) * 4;
      }
    }
    qa2->leaf_oracle = K ? 1 : 0;  // the trace follows the constants
    TRY(launch_query_initial(*qa2, B, n_or, st));
    TRY(launch_query_layers(*ql, B, L.n_layers, st));
    if (q_direct) {
      TRY(w.wait());
      for (uint32_t b = 0; b < B; b++) std::memcpy(P[b] + L.queries, w.pinned + b * q_words, q_words * 8);
      if (first_trace_leaf) std::memcpy(first_trace_leaf, w.pinned + B * q_words, 4 * B * 8);
    } else {
      for (uint32_t b = 0; b < B; b++) TRY(w.d2h(P[b] + L.queries, d_q + b * q_words, q_words));
      if (first_trace_leaf) TRY(w.d2h(first_trace_leaf, d_q + B * q_words, 4 * B));
    }
  }
  return BP_OK;
}

}  // namespace bpg
