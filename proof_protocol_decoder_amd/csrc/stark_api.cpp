// stark_api.cpp -- C ABI entry for one synthetic-AIR table proof (include/bpg.h, L0.5).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include "prover.hpp"

using namespace bpg;

namespace {
constexpr uint32_t LONE_PI_LEN = 4;
void lone_public_input_list(uint64_t seed, uint64_t out[LONE_PI_LEN]) {
  for (uint64_t j = 0; j < LONE_PI_LEN; j++) {
    uint64_t z = (seed ^ ((0x50 + j) << 32)) + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    out[j] = gl::canon(z ^ (z >> 31));
  }
}
}  // namespace
namespace {
// One parked worker (stream + arena) per device: a 2^20 x 2432 table needs a ~100 GB arena, and
// hipFree + hipMalloc of that much costs seconds -- several times the proof itself.  A call takes
// the parked worker if its arena is large enough (concurrent calls simply make their own), and parks
// its worker again when the slot is free.  bp_release_cached_memory() empties the slots.
std::mutex g_park_mu;
std::vector<Worker*> g_parked;  // index = device

void park_worker(Worker* w);

Worker* take_worker(int device, size_t bytes, int* rc) {
  Worker* w = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_park_mu);
    if (device >= 0 && (size_t)device < g_parked.size() && g_parked[device]) {
      w = g_parked[device];
      g_parked[device] = nullptr;
    }
  }
  if (w && w->arena.capacity() >= bytes) {
    if (hipSetDevice(device) != hipSuccess) {
      *rc = fail(BP_ERR_DEVICE, "hipSetDevice(%d) failed", device);
      park_worker(w);
      return nullptr;
    }
    w->arena.release(0);
    w->abort_flag = nullptr;
    *rc = BP_OK;
    return w;
  }
  if (w) {
    w->destroy();
    delete w;
  }
  w = new Worker();
  *rc = w->init(device, bytes);
  if (*rc) {
    w->destroy();
    delete w;
    return nullptr;
  }
  return w;
}
void park_worker(Worker* w) {
  {
    std::lock_guard<std::mutex> lk(g_park_mu);
    if (w->device >= 0) {
      if (g_parked.size() <= (size_t)w->device) g_parked.resize(w->device + 1, nullptr);
      if (!g_parked[w->device]) {
        g_parked[w->device] = w;
        return;
      }
    }
  }
  w->destroy();
  delete w;
}
}  // namespace

extern "C" {

void bp_free_buffer(uint8_t* buf) { std::free(buf); }
void bp_tune_k5_spread(int on) { bpg::g_k5_spread_all.store(on != 0); }
void bp_tune_host_wait(int mode) { bpg::tune_host_wait(mode); }
void bp_tune_host_poseidon(int mode) { bpg::tune_host_poseidon(mode); }
// Host only: the transcript's CPU permutation (prover.cpp, poseidon_host) over n states of 12 words, in place.
int bp_debug_poseidon_host(uint64_t* states, size_t n) try {
  if (!states && n) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_debug_poseidon_host: null states");
  for (size_t i = 0; i < n; i++) bpg::poseidon_host(states + 12 * i);
  return BP_OK;
}
BPG_ABI_CATCH("bp_debug_poseidon_host")

void bp_release_cached_memory(void) {
  std::vector<Worker*> all;
  {
    std::lock_guard<std::mutex> lk(g_park_mu);
    all.swap(g_parked);
  }
  for (Worker* w : all)
    if (w) {
      w->destroy();
      delete w;
    }
}

int bp_stark_prove_air(uint32_t air_id, const bp_stark_cfg* cfg, uint64_t seed, uint64_t const_seed, int device,
                       uint8_t** out, size_t* out_len) try {
  if (!cfg || !out || !out_len) return fail(BP_ERR_INVALID_INPUT, "bp_stark_prove_air: null argument");
  StarkCfg c{cfg->log_n, cfg->n_cols, cfg->n_const, cfg->deg_pow, cfg->rate_bits, cfg->cap_height,
             cfg->num_queries, cfg->pow_bits, cfg->arity_bits, cfg->final_poly_bits, air_id};
  int rc = check_cfg(c);
  if (rc) return rc;
  const uint64_t N = (uint64_t)1 << c.log_n, M = N << c.rate_bits;
  // generous one-shot arena: every oracle (values + coeffs + LDE) plus temporaries
  const size_t cols = (size_t)c.n_cols + c.n_const + c.n_cols / 8 + (2u << c.rate_bits) + 64;
  size_t bytes = cols * (2 * N + M) * 8 + (size_t)80 * M * 8 + ((size_t)c.n_cols / 32 + 64) * 8 * N * 8 + (64u << 20);
  Worker* wp = take_worker(device, bytes, &rc);
  if (!wp) return rc;
  Worker& w = *wp;
  // parked again (or freed) on every way out, an exception included: the worker holds the arena
  struct Parker {
    Worker* w;
    ~Parker() { park_worker(w); }
  } parker{wp};
  auto body = [&]() -> int {
    Challenger ch;
    Committed consts, trace;
    uint64_t* d_consts = nullptr;
    if (c.n_const) {
      d_consts = w.arena.alloc_words((size_t)c.n_const * N);
      if (!d_consts) return fail(BP_ERR_DEVICE, "arena exhausted");
      int r2 = c.air_id == air::PLONK ? launch_plonk_constants(d_consts, c.log_n, const_seed, air::plonk::Layout{LONE_PI_LEN, 0, 0, 0}, w.stream)
                                      : launch_synth_constants(d_consts, c.log_n, c.n_const, const_seed, w.stream);
      if (r2) return r2;
      if ((r2 = commit(w, d_consts, c.n_const, c.log_n, c.rate_bits, c.cap_height, false, &consts))) return r2;
      ch.observe(consts.cap.data(), consts.cap.size());
    }
    uint64_t* d_trace = w.arena.alloc_words((size_t)c.n_cols * N);
    if (!d_trace) return fail(BP_ERR_DEVICE, "arena exhausted");
    int r2 = c.air_id == air::KECCAK_F ? launch_keccak_trace(d_trace, nullptr, c.log_n, seed, w.stream)
             : c.air_id == air::LOGIC  ? launch_logic_trace(d_trace, nullptr, c.log_n, seed, w.stream)
             : c.air_id == air::MEMORY ? launch_memory_trace(d_trace, nullptr, c.log_n, seed, w.stream)
             : c.air_id == air::ARITHMETIC ? launch_arithmetic_trace(d_trace, nullptr, c.log_n, seed, w.stream)
             : c.air_id == air::BYTE_PACKING ? launch_byte_packing_trace(d_trace, nullptr, c.log_n, seed, w.stream)
             : c.air_id == air::KECCAK_SPONGE ? launch_keccak_sponge_trace(d_trace, nullptr, c.log_n, seed, w.stream)
             : c.air_id == air::ARITHMETIC_MUL ? launch_arithmetic_mul_trace(d_trace, nullptr, c.log_n, seed, w.stream)
             : c.air_id == air::PLONK ? BP_OK
                                       : launch_synth_trace(d_trace, d_consts, c.log_n, c.n_cols, c.n_const, c.deg_pow, seed, w.stream);
    Ctl ctl;
    if (c.air_id == air::PLONK) {  // the public-input list of a lone table proof follows from the seed (lone_public_input_list)
      uint64_t pi[LONE_PI_LEN];
      lone_public_input_list(seed, pi);
      std::vector<uint64_t> rows;
      poseidon_hash_rows(pi, LONE_PI_LEN, &rows, ctl.pub);   // the circuit hashes the list in its hash rows: pub = that hash
      std::memcpy(w.hash_rows, rows.data(), rows.size() * 8);
      PlonkTraceArgs pa{d_trace, d_consts, seed, {ctl.pub[0], ctl.pub[1], ctl.pub[2], ctl.pub[3]}, w.hash_rows_dev, 1};
      r2 = launch_plonk_trace(&pa, 1, c.log_n, w.stream);
    }
    if (r2) return r2;
    if ((r2 = commit(w, d_trace, c.n_cols, c.log_n, c.rate_bits, c.cap_height, false, &trace))) return r2;
    ch.observe(trace.cap.data(), trace.cap.size());
    for (int i = 0; i < 4; i++) ctl.v[i] = ch.challenge();
    std::vector<uint64_t> proof;
    if ((r2 = stark_prove(w, c, c.n_const ? &consts : nullptr, trace, d_trace, ctl, ch, proof))) return r2;
    *out_len = proof.size() * 8;
    *out = static_cast<uint8_t*>(std::malloc(*out_len));
    if (!*out) return fail(BP_ERR_DEVICE, "host allocation failed");
    std::memcpy(*out, proof.data(), *out_len);
    return BP_OK;
  };
  rc = body();
  // nothing of this call is still in flight when the worker is parked (a failed call keeps its own message)
  if (rc) (void)hipStreamSynchronize(w.stream);
  else rc = w.wait();
  return rc;
}
BPG_ABI_CATCH("bp_stark_prove_air")

int bp_stark_prove_synthetic(const bp_stark_cfg* cfg, uint64_t seed, uint64_t const_seed, int device,
                             uint8_t** out, size_t* out_len) {
  return bp_stark_prove_air(air::SYNTHETIC, cfg, seed, const_seed, device, out, out_len);
}

// The CPU verifier for one table proof of bp_stark_prove_air (same transcript prologue: constants cap if any, trace
// cap, four CTL challenges).  Host only: runs without a GPU.
// A lone AIR-8 table proof made from `seed` has the public-input LIST splitmix64(seed ^ ((0x50 + j) << 32)) mod p, j < 4
// (in a transaction the list is child digests and public values); the four public inputs bound to its first row are the
// hash of that list, which the circuit computes in its hash rows and the verifier computes for itself.
void bp_stark_public_input_list(uint64_t seed, uint64_t out[4]) { lone_public_input_list(seed, out); }
void bp_stark_public_inputs(uint64_t seed, uint64_t out[4]) {
  uint64_t pi[LONE_PI_LEN];
  lone_public_input_list(seed, pi);
  hash_no_pad_host(pi, LONE_PI_LEN, out);
}

int bp_stark_verify_air(uint32_t air_id, const bp_stark_cfg* cfg, const uint64_t* const_cap, const uint8_t* proof,
                        size_t len) {
  return bp_stark_verify_air_pub(air_id, cfg, const_cap, nullptr, proof, len);
}
int bp_stark_verify_air_pub(uint32_t air_id, const bp_stark_cfg* cfg, const uint64_t* const_cap, const uint64_t* pub,
                            const uint8_t* proof, size_t len) try {
  if (!cfg || !proof) return fail(BP_ERR_INVALID_INPUT, "bp_stark_verify_air: null argument");
  StarkCfg c{cfg->log_n, cfg->n_cols, cfg->n_const, cfg->deg_pow, cfg->rate_bits, cfg->cap_height,
             cfg->num_queries, cfg->pow_bits, cfg->arity_bits, cfg->final_poly_bits, air_id};
  int rc = check_cfg(c);
  if (rc) return rc;
  if (c.n_const && !const_cap) return fail(BP_ERR_INVALID_INPUT, "bp_stark_verify_air: the table has constant columns: pass their cap");
  const ProofLayout L = proof_layout(c);
  if (len != L.total * 8) return fail(BP_ERR_VERIFY, "proof has %zu bytes, expected %zu", len, L.total * 8);
  std::vector<uint64_t> w(L.total);
  std::memcpy(w.data(), proof, len);
  Challenger ch;
  if (c.n_const) ch.observe(const_cap, L.cap_words);
  ch.observe(w.data() + L.trace_cap, L.cap_words);
  Ctl ctl;
  for (int i = 0; i < 4; i++) ctl.v[i] = ch.challenge();
  if (pub)
    for (int i = 0; i < 4; i++) {
      if (pub[i] >= gl::P) return fail(BP_ERR_INVALID_INPUT, "non-canonical public input");
      ctl.pub[i] = pub[i];
    }
  return stark_verify(c, c.n_const ? const_cap : nullptr, ctl, ch, w.data(), w.size());
}
BPG_ABI_CATCH("bp_stark_verify_air_pub")

// ---- the AIR registry (air.hpp) --------------------------------------------------------------------------

uint32_t bp_air_count(void) { return air::COUNT; }

int bp_air_describe(uint32_t air_id, uint32_t n_cols, uint32_t n_const, uint32_t deg_pow, bp_air_desc* out) try {
  const air::Info* ai = air::info(air_id);
  if (!ai || !out) return fail(BP_ERR_INVALID_INPUT, "bp_air_describe: unknown air_id %u or null output", air_id);
  std::memset(out, 0, sizeof(*out));
  out->air_id = air_id;
  std::strncpy(out->name, ai->name, sizeof(out->name) - 1);
  out->fixed_n_cols = ai->n_cols;
  out->n_const_max = ai->n_const_max;
  const uint32_t C = ai->n_cols ? ai->n_cols : n_cols, dp = ai->n_cols ? (ai->degree > 3 ? 3 : 1) : (deg_pow ? deg_pow : 1);
  const air::Shape shape{air_id, C, ai->n_cols ? ai->n_const_max : n_const, dp};
  out->degree = ai->n_cols ? ai->degree : ai->degree * dp;
  out->n_cols = C;
  out->n_aux = air::ctl::n_aux(shape);
  out->n_air_constraints = air::n_constraints(shape);
  out->n_ctl_constraints = air::ctl::n_constraints(shape);
  out->n_units = air::n_units(shape);
  // families: (first index, count, kind, degree); kinds: 0 all rows, 1 transition, 2 first row, 3 last row
  uint32_t n = 0;
  auto fam = [&](uint32_t first, uint32_t count, uint32_t kind, uint32_t degree) {
    if (n < 24) out->families[n++] = bp_air_family{first, count, kind, degree};
  };
  if (air_id == air::KECCAK_F) {
    namespace kk = air::keccak;
    fam(kk::F0, 24, 2, 1); fam(kk::F1, 24, 1, 1); fam(kk::F2, 1984, 0, 2); fam(kk::F3, 320, 0, 3); fam(kk::F4, 320, 0, 3);
    fam(kk::F5, 50, 0, 3); fam(kk::F6, 50, 0, 3); fam(kk::F7, 2, 0, 1); fam(kk::F8, 2, 0, 2); fam(kk::F9, 50, 1, 2);
  } else if (air_id == air::LOGIC) {
    namespace lg = air::logic;
    fam(lg::L0, 3, 0, 2); fam(lg::L1, 1, 0, 2); fam(lg::L2, 512, 0, 2); fam(lg::L3, 8, 0, 3);
  } else if (air_id == air::MEMORY) {
    namespace mm = air::memory;
    fam(mm::M0, 1, 0, 2); fam(mm::M1, 1, 0, 2); fam(mm::M2, 32, 0, 2); fam(mm::M3, 1, 1, 2); fam(mm::M4, 1, 1, 2);
    fam(mm::M5, 8, 1, 3); fam(mm::M6, 8, 1, 3); fam(mm::M7, 8, 2, 2);
  } else if (air_id == air::ARITHMETIC) {
    namespace ar = air::arithmetic;
    fam(ar::A0, 4, 0, 2); fam(ar::A1, 1, 0, 2); fam(ar::A2, 256, 0, 2); fam(ar::A3, 16, 0, 2); fam(ar::A4, 16, 0, 2);
    fam(ar::A5, 1, 0, 2);
  } else if (air_id == air::BYTE_PACKING) {
    namespace bk = air::byte_packing;
    fam(bk::P0, 1, 0, 2); fam(bk::P1, 32, 0, 2); fam(bk::P2, 1, 0, 2); fam(bk::P3, 256, 0, 2); fam(bk::P4, 32, 0, 2);
    fam(bk::P5, 8, 0, 2);
  } else if (air_id == air::KECCAK_SPONGE) {
    namespace sp = air::keccak_sponge;
    fam(sp::K0, 2, 0, 2); fam(sp::K1, 1, 0, 2); fam(sp::K2, 136, 0, 2); fam(sp::K3, 1, 0, 1); fam(sp::K4, 1088, 0, 2);
    fam(sp::K5, 1088, 0, 2); fam(sp::K6, 136, 0, 2); fam(sp::K7, 34, 0, 2); fam(sp::K8, 50, 1, 2); fam(sp::K9, 50, 2, 1);
    fam(sp::K10, 1, 1, 2);
  } else if (air_id == air::PLONK) {
    namespace pk = air::plonk;
    fam(pk::G0, 20, 0, 4); fam(pk::G1, 44, 0, 3); fam(pk::G2, 22, 0, 2); fam(pk::G3, 4, 2, 1); fam(pk::G4, 118, 0, 8); fam(pk::G5, 5, 0, 3);
  } else if (air_id == air::ARITHMETIC_MUL) {
    namespace am = air::arithmetic_mul;
    fam(am::U0, 1, 0, 2); fam(am::U1, 256, 0, 2); fam(am::U2, 256, 0, 2); fam(am::U3, 672, 0, 2); fam(am::U4, 32, 0, 3);
    fam(am::U5, 1, 0, 1);
  } else {
    // interleaved per group of four columns: 3g all rows, 3g + 1 transition, 3g + 2 first row
    fam(0, C / 4, 0, 2); fam(1, C / 4, 1, 3 * dp); fam(2, C / 4, 2, 1);
  }
  // the table's lookups (air::ctl), in list order after the AIR's own constraints
  const uint32_t b = out->n_air_constraints;
  if (air_id == air::SYNTHETIC) {
    fam(b, C / 8, 1, 2); fam(b + 1, C / 8, 3, 1);  // running products, interleaved 2k (transition), 2k + 1 (last row)
  } else if (air_id == air::PLONK) {  // the copy constraints: per challenge set ten chunk relations and Z(first) = 1
    for (uint32_t c = 0; c < 2; c++) { fam(b + 11 * c, 10, 0, 9); fam(b + 11 * c + 10, 1, 2, 1); }
  } else {
    uint32_t i = b;
    if (air_id == air::KECCAK_F) {
      fam(i, 2, 0, 2); i += 2;           // the filter g: a bit, set on last-round rows only
      for (int c = 0; c < 2; c++) {      // h_c: the compressed input, fixed on first-round rows, carried along
        fam(i, 1, 0, 2); fam(i + 1, 1, 1, 2); i += 2;
      }
    }
    if (air_id == air::MEMORY || air_id == air::LOGIC) { fam(i, 1, 0, 2); i += 1; }  // the filter g of the looked table: a bit
    const uint32_t n_products = out->n_aux - air::ctl::first_product(air_id);
    if (n_products > 4) {  // (the sponge table's twelve) interleaved like the synthetic tables': 2k transition, 2k + 1 last row
      fam(i, n_products, 1, 3); fam(i + 1, n_products, 3, 2);
    } else {
      for (uint32_t k = 0; k < n_products; k++) {  // filtered running products
        fam(i, 1, 1, 3); fam(i + 1, 1, 3, 2); i += 2;
      }
    }
  }
  out->n_families = n;
  return BP_OK;
}
BPG_ABI_CATCH("bp_air_describe")

int bp_keccak_trace(const uint64_t* d_inputs, uint64_t seed, uint32_t log_n, uint64_t* d_trace_out, void* stream) try {
  if (!d_trace_out) return fail(BP_ERR_INVALID_INPUT, "bp_keccak_trace: null output");
  if (log_n < 4 || log_n > 26) return fail(BP_ERR_INVALID_INPUT, "bp_keccak_trace: log_n out of range");
  return launch_keccak_trace(d_trace_out, d_inputs, log_n, seed, as_stream(stream));
}
BPG_ABI_CATCH("bp_keccak_trace")

int bp_logic_trace(const uint64_t* d_inputs, uint64_t seed, uint32_t log_n, uint64_t* d_trace_out, void* stream) try {
  if (!d_trace_out) return fail(BP_ERR_INVALID_INPUT, "bp_logic_trace: null output");
  if (log_n < 4 || log_n > 26) return fail(BP_ERR_INVALID_INPUT, "bp_logic_trace: log_n out of range");
  return launch_logic_trace(d_trace_out, d_inputs, log_n, seed, as_stream(stream));
}
BPG_ABI_CATCH("bp_logic_trace")

int bp_memory_trace(const uint64_t* d_inputs, uint64_t seed, uint32_t log_n, uint64_t* d_trace_out, void* stream) try {
  if (!d_trace_out) return fail(BP_ERR_INVALID_INPUT, "bp_memory_trace: null output");
  if (log_n < 4 || log_n > 26) return fail(BP_ERR_INVALID_INPUT, "bp_memory_trace: log_n out of range");
  return launch_memory_trace(d_trace_out, d_inputs, log_n, seed, as_stream(stream));
}
BPG_ABI_CATCH("bp_memory_trace")

int bp_arithmetic_trace(const uint64_t* d_inputs, uint64_t seed, uint32_t log_n, uint64_t* d_trace_out, void* stream) try {
  if (!d_trace_out) return fail(BP_ERR_INVALID_INPUT, "bp_arithmetic_trace: null output");
  if (log_n < 4 || log_n > 26) return fail(BP_ERR_INVALID_INPUT, "bp_arithmetic_trace: log_n out of range");
  return launch_arithmetic_trace(d_trace_out, d_inputs, log_n, seed, as_stream(stream));
}
BPG_ABI_CATCH("bp_arithmetic_trace")

int bp_byte_packing_trace(const uint64_t* d_inputs, uint64_t seed, uint32_t log_n, uint64_t* d_trace_out, void* stream) try {
  if (!d_trace_out) return fail(BP_ERR_INVALID_INPUT, "bp_byte_packing_trace: null output");
  if (log_n < 4 || log_n > 26) return fail(BP_ERR_INVALID_INPUT, "bp_byte_packing_trace: log_n out of range");
  return launch_byte_packing_trace(d_trace_out, d_inputs, log_n, seed, as_stream(stream));
}
BPG_ABI_CATCH("bp_byte_packing_trace")

int bp_keccak_sponge_trace(const uint64_t* d_inputs, uint64_t seed, uint32_t log_n, uint64_t* d_trace_out, void* stream) try {
  if (!d_trace_out) return fail(BP_ERR_INVALID_INPUT, "bp_keccak_sponge_trace: null output");
  if (log_n < 4 || log_n > 26) return fail(BP_ERR_INVALID_INPUT, "bp_keccak_sponge_trace: log_n out of range");
  return launch_keccak_sponge_trace(d_trace_out, d_inputs, log_n, seed, as_stream(stream));
}
BPG_ABI_CATCH("bp_keccak_sponge_trace")

int bp_arithmetic_mul_trace(const uint64_t* d_inputs, uint64_t seed, uint32_t log_n, uint64_t* d_trace_out, void* stream) try {
  if (!d_trace_out) return fail(BP_ERR_INVALID_INPUT, "bp_arithmetic_mul_trace: null output");
  if (log_n < 4 || log_n > 26) return fail(BP_ERR_INVALID_INPUT, "bp_arithmetic_mul_trace: log_n out of range");
  return launch_arithmetic_mul_trace(d_trace_out, d_inputs, log_n, seed, as_stream(stream));
}
BPG_ABI_CATCH("bp_arithmetic_mul_trace")

// AIR 8: the preprocessed constants of the fixed PLONK-shaped circuit (85 columns: selectors, gate constants drawn from
// `seed`, the hash-row selector, sigmas) for a circuit that hashes a public-input list of pi_len words, and its witness
// (135 wires; free wires drawn from `seed`; the list pi is hashed in the hash rows, the hash lands in row 0).
static int plonk_layout_of(const bp_plonk_layout* l, uint32_t log_n, air::plonk::Layout* out) {
  if (!l) return fail(BP_ERR_INVALID_INPUT, "null plonk layout");
  *out = air::plonk::Layout{l->pi_len, l->n_paths, l->path_depth, l->path_pi0, l->leaf_len};
  if (!air::plonk::layout_ok(*out, 1u << log_n))
    return fail(BP_ERR_INVALID_INPUT, "plonk layout: a list of 1..%u words, at most %u Merkle rows (n_paths x path_depth), the paths' words "
                "(8 per path from path_pi0) inside the list, and room for one arithmetic group in 2^%u rows", air::plonk::MAX_PI,
                air::plonk::MERKLE_ROWS_MAX, log_n);
  return BP_OK;
}
int bp_plonk_constants(uint64_t seed, uint32_t log_n, const bp_plonk_layout* layout, uint64_t* d_consts_out, void* stream) try {
  if (!d_consts_out) return fail(BP_ERR_INVALID_INPUT, "bp_plonk_constants: null output");
  if (log_n < 4 || log_n > 26) return fail(BP_ERR_INVALID_INPUT, "bp_plonk_constants: log_n out of range");
  air::plonk::Layout lay;
  int rc = plonk_layout_of(layout, log_n, &lay);
  if (rc) return rc;
  if ((rc = init_ntt_kernels())) return rc;
  return launch_plonk_constants(d_consts_out, log_n, seed, lay, as_stream(stream));
}
BPG_ABI_CATCH("bp_plonk_constants")
int bp_plonk_trace(const uint64_t* d_consts, uint64_t seed, const uint64_t* pi, const bp_plonk_layout* layout, const uint64_t* paths,
                   uint32_t log_n, uint64_t* d_trace_out, void* stream) try {
  if (!d_consts || !pi || !d_trace_out) return fail(BP_ERR_INVALID_INPUT, "bp_plonk_trace: null argument");
  if (log_n < 4 || log_n > 26) return fail(BP_ERR_INVALID_INPUT, "bp_plonk_trace: log_n out of range");
  air::plonk::Layout lay;
  int rc = plonk_layout_of(layout, log_n, &lay);
  if (rc) return rc;
  if (lay.n_paths && !paths) return fail(BP_ERR_INVALID_INPUT, "bp_plonk_trace: the layout walks %u Merkle paths: their witness is missing", lay.n_paths);
  for (uint32_t j = 0; j < lay.pi_len; j++) if (pi[j] >= gl::P) return fail(BP_ERR_INVALID_INPUT, "bp_plonk_trace: non-canonical public input");
  std::vector<uint64_t> rows, all((size_t)(air::plonk::HASH_ROWS_MAX + air::plonk::merkle_rows(lay) + air::plonk::leaf_rows(lay)) * air::plonk::H_WIRES, 0);
  uint64_t pub[4];
  poseidon_hash_rows(pi, lay.pi_len, &rows, pub);
  std::memcpy(all.data(), rows.data(), rows.size() * 8);
  const uint32_t n_list_rows = (uint32_t)(rows.size() / air::plonk::H_WIRES);
  const size_t path_words = 1 + 4 * (size_t)lay.depth + lay.leaf_len;
  for (uint32_t p = 0; p < lay.n_paths; p++) {
    const uint64_t* pw = paths + p * path_words;
    for (size_t j = 1; j < path_words; j++) if (pw[j] >= gl::P) return fail(BP_ERR_INVALID_INPUT, "bp_plonk_trace: non-canonical sibling word");
    uint64_t root[4];
    poseidon_merkle_rows(pi + lay.path_pi0 + 8 * p, pw[0], pw + 1, lay.depth,
                         all.data() + (size_t)(air::plonk::HASH_ROWS_MAX + p * lay.depth) * air::plonk::H_WIRES, root);
    if (lay.leaf_len) {
      uint64_t digest[4];
      poseidon_hash_rows(pw + 1 + 4 * (size_t)lay.depth, lay.leaf_len, &rows, digest);
      std::memcpy(all.data() + (size_t)(air::plonk::HASH_ROWS_MAX + air::plonk::merkle_rows(lay) + p * air::plonk::hash_rows(lay.leaf_len)) * air::plonk::H_WIRES,
                  rows.data(), rows.size() * 8);
    }
  }
  // (a test / integration entry: the rows go up with a blocking copy into a buffer of their own)
  uint64_t* d_rows = nullptr;
  BPG_HIP(hipMalloc(reinterpret_cast<void**>(&d_rows), all.size() * 8));
  struct Free { uint64_t* p; ~Free() { (void)hipFree(p); } } guard{d_rows};
  BPG_HIP(hipMemcpy(d_rows, all.data(), all.size() * 8, hipMemcpyHostToDevice));
  PlonkTraceArgs a{d_trace_out, d_consts, seed, {pub[0], pub[1], pub[2], pub[3]}, d_rows, n_list_rows,
                   air::plonk::merkle_rows(lay) + air::plonk::leaf_rows(lay), air::plonk::arith_row0(lay)};
  rc = launch_plonk_trace(&a, 1, log_n, as_stream(stream));
  if (rc) return rc;
  BPG_HIP(hipStreamSynchronize(as_stream(stream)));
  return BP_OK;
}
BPG_ABI_CATCH("bp_plonk_trace")

// ---- L0: the remaining per-stage entry points of SURVEY.md section 8(b) -------------------------------

static int quot_cfg(uint32_t air_id, const bp_stark_cfg* shape, StarkCfg* c) {
  if (!shape) return fail(BP_ERR_INVALID_INPUT, "null shape");
  *c = StarkCfg{shape->log_n, shape->n_cols, shape->n_const, shape->deg_pow, shape->rate_bits, shape->cap_height,
                shape->num_queries, shape->pow_bits, shape->arity_bits, shape->final_poly_bits, air_id};
  return check_cfg(*c);
}

uint64_t bp_quotient_scratch_words(uint32_t air_id, const bp_stark_cfg* shape) {
  StarkCfg c;
  if (quot_cfg(air_id, shape, &c)) return 0;
  // The spreading of the units over workgroup rows follows the device's load, which bp_generate_* calls on other
  // threads change at any moment: the scratch is sized for the larger of the two forms (the unloaded one spreads
  // furthest), so a launch that finds another load state than this call did still fits.
  QuotArgs qa{};
  Ctl ctl{};
  if (init_ntt_kernels()) return 0;
  uint64_t words = 0;
  for (int loaded = 0; loaded < 2; loaded++) {
    if (quotient_args(c, ctl, 1, 1, &qa, nullptr, loaded)) return 0;
    words = std::max<uint64_t>(words, 2 * (uint64_t)qa.n_constraints + 48 + quotient_partial_words(qa));
  }
  return words;
}

int bp_quotient_eval(uint32_t air_id, const bp_stark_cfg* shape, const uint64_t* d_trace_lde, const uint64_t* d_aux_lde,
                     const uint64_t* d_const_lde, const uint64_t ctl_in[4], const uint64_t alphas[2],
                     uint64_t* d_scratch, uint64_t* d_qvals_out, void* stream) try {
  StarkCfg c;
  int rc = quot_cfg(air_id, shape, &c);
  if (rc) return rc;
  if (!d_trace_lde || !d_aux_lde || (c.n_const && !d_const_lde) || !ctl_in || !alphas || !d_scratch || !d_qvals_out)
    return fail(BP_ERR_INVALID_INPUT, "bp_quotient_eval: null argument");
  for (int i = 0; i < 4; i++)
    if (ctl_in[i] >= gl::P) return fail(BP_ERR_INVALID_INPUT, "bp_quotient_eval: non-canonical challenge");
  if (alphas[0] >= gl::P || alphas[1] >= gl::P) return fail(BP_ERR_INVALID_INPUT, "bp_quotient_eval: non-canonical alpha");
  if ((rc = init_ntt_kernels())) return rc;
  Ctl ctl;
  for (int i = 0; i < 4; i++) ctl.v[i] = ctl_in[i];
  QuotArgs qa{};
  QuotCoset coset{};
  qa.trace_lde = d_trace_lde; qa.aux_lde = d_aux_lde; qa.const_lde = c.n_const ? d_const_lde : nullptr;
  if ((rc = quotient_args(c, ctl, alphas[0], alphas[1], &qa, &coset))) return rc;
  qa.apow = d_scratch; qa.partial = d_scratch + 2 * (size_t)qa.n_constraints + 48; qa.qvals = d_qvals_out;
  return launch_quotient(qa, coset, as_stream(stream));
}
BPG_ABI_CATCH("bp_quotient_eval")

int bp_fri_fold(const uint64_t* d_values, uint32_t log_nl, uint32_t rate_bits, uint32_t arity_bits, uint64_t shift,
                const uint64_t beta[2], uint64_t* d_out, void* stream) try {
  if (!d_values || !d_out || !beta) return fail(BP_ERR_INVALID_INPUT, "bp_fri_fold: null argument");
  if (shift == 0 || shift >= gl::P || beta[0] >= gl::P || beta[1] >= gl::P)
    return fail(BP_ERR_INVALID_INPUT, "bp_fri_fold: non-canonical field element");
  int rc = init_ntt_kernels();
  if (rc) return rc;
  FriLayerArgs fa{};
  if ((rc = fri_layer_args(log_nl, rate_bits, arity_bits, shift, &fa))) return rc;
  fa.values = d_values; fa.out = d_out; fa.digests = nullptr;
  fa.beta = gl::Ext{beta[0], beta[1]};
  return launch_fri_fold(fa, as_stream(stream));
}
BPG_ABI_CATCH("bp_fri_fold")

int bp_pow_grind(const uint64_t state[12], uint32_t pos, uint32_t bits, uint64_t* nonce_out, void* stream) try {
  if (!state || !nonce_out) return fail(BP_ERR_INVALID_INPUT, "bp_pow_grind: null argument");
  if (pos >= 8 || bits == 0 || bits > 40) return fail(BP_ERR_INVALID_INPUT, "bp_pow_grind: pos must be a rate word, bits in 1..40");
  hipStream_t st = as_stream(stream);
  unsigned long long* d_res = nullptr;
  BPG_HIP(hipMalloc(reinterpret_cast<void**>(&d_res), 8));
  PowArgs pa{};
  for (int i = 0; i < 12; i++) pa.state[i] = state[i];
  pa.pos = pos; pa.bits = bits;
  int rc = BP_OK;
  unsigned long long res = ~0ULL;
  if (hipMemsetAsync(d_res, 0xFF, 8, st) != hipSuccess) rc = fail(BP_ERR_DEVICE, "hipMemsetAsync failed");
  uint32_t batch = (uint32_t)std::min<uint64_t>(1u << 20, std::max<uint64_t>(1u << 12, (uint64_t)2 << bits));
  for (uint64_t base = 0; rc == BP_OK; base += batch, batch = std::min<uint32_t>(1u << 20, batch * 2)) {
    pa.base = base;
    if ((rc = launch_pow(pa, batch, d_res, st))) break;
    if (hipMemcpyAsync(&res, d_res, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
      rc = fail(BP_ERR_DEVICE, "bp_pow_grind: copy back failed");
      break;
    }
    if (res != ~0ULL) break;
    if (base > ((uint64_t)1 << 44)) rc = fail(BP_ERR_DEVICE, "proof of work search exhausted");
  }
  (void)hipFree(d_res);
  if (rc == BP_OK) *nonce_out = res;
  return rc;
}
BPG_ABI_CATCH("bp_pow_grind")

int bp_openings(const uint64_t* d_coeffs, uint64_t stride, uint32_t log_n, uint32_t n_cols, const uint64_t z0[2],
                const uint64_t z1[2], uint64_t* d_pw_scratch, uint64_t* d_out, void* stream) try {
  if (!n_cols) return BP_OK;
  if (!d_coeffs || !z0 || !d_pw_scratch || !d_out) return fail(BP_ERR_INVALID_INPUT, "bp_openings: null argument");
  if (log_n > 30 || stride < ((uint64_t)1 << log_n)) return fail(BP_ERR_INVALID_INPUT, "bp_openings: bad shape");
  const uint64_t* zz[2] = {z0, z1 ? z1 : z0};
  for (int k = 0; k < 2; k++)
    if (zz[k][0] >= gl::P || zz[k][1] >= gl::P) return fail(BP_ERR_INVALID_INPUT, "bp_openings: non-canonical point");
  hipStream_t st = as_stream(stream);
  const uint32_t n_points = z1 ? 2 : 1;
  int rc = launch_power_vectors(d_pw_scratch, log_n, gl::Ext{z0[0], z0[1]}, gl::Ext{zz[1][0], zz[1][1]}, n_points, st);
  if (rc) return rc;
  return launch_openings(d_coeffs, stride, log_n, n_cols, d_pw_scratch, n_points, d_out, st);
}
BPG_ABI_CATCH("bp_openings")

}  // extern "C"
