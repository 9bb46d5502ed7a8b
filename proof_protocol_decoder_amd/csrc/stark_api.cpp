// stark_api.cpp -- C ABI entry for one synthetic-AIR table proof (include/bpg.h, L0.5).
#include <cstdlib>
#include <mutex>
#include <vector>
#include "prover.hpp"

using namespace bpg;

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

int bp_stark_prove_synthetic(const bp_stark_cfg* cfg, uint64_t seed, uint64_t const_seed, int device,
                             uint8_t** out, size_t* out_len) {
  if (!cfg || !out || !out_len) return fail(BP_ERR_INVALID_INPUT, "bp_stark_prove_synthetic: null argument");
  StarkCfg c{cfg->log_n, cfg->n_cols, cfg->n_const, cfg->deg_pow, cfg->rate_bits, cfg->cap_height,
             cfg->num_queries, cfg->pow_bits, cfg->arity_bits, cfg->final_poly_bits};
  int rc = check_cfg(c);
  if (rc) return rc;
  const uint64_t N = (uint64_t)1 << c.log_n, M = N << c.rate_bits;
  // generous one-shot arena: every oracle (values + coeffs + LDE) plus temporaries
  const size_t cols = (size_t)c.n_cols + c.n_const + c.n_cols / 8 + (2u << c.rate_bits) + 64;
  size_t bytes = cols * (2 * N + M) * 8 + (size_t)80 * M * 8 + ((size_t)c.n_cols / 32 + 64) * 8 * N * 8 + (64u << 20);
  Worker* wp = take_worker(device, bytes, &rc);
  if (!wp) return rc;
  Worker& w = *wp;
  auto body = [&]() -> int {
    Challenger ch;
    Committed consts, trace;
    uint64_t* d_consts = nullptr;
    if (c.n_const) {
      d_consts = w.arena.alloc_words((size_t)c.n_const * N);
      if (!d_consts) return fail(BP_ERR_DEVICE, "arena exhausted");
      int r2 = launch_synth_constants(d_consts, c.log_n, c.n_const, const_seed, w.stream);
      if (r2) return r2;
      if ((r2 = commit(w, d_consts, c.n_const, c.log_n, c.rate_bits, c.cap_height, false, &consts))) return r2;
      ch.observe(consts.cap.data(), consts.cap.size());
    }
    uint64_t* d_trace = w.arena.alloc_words((size_t)c.n_cols * N);
    if (!d_trace) return fail(BP_ERR_DEVICE, "arena exhausted");
    int r2 = launch_synth_trace(d_trace, d_consts, c.log_n, c.n_cols, c.n_const, c.deg_pow, seed, w.stream);
    if (r2) return r2;
    if ((r2 = commit(w, d_trace, c.n_cols, c.log_n, c.rate_bits, c.cap_height, false, &trace))) return r2;
    ch.observe(trace.cap.data(), trace.cap.size());
    Ctl ctl;
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
  if (rc == BP_OK) rc = w.wait();  // nothing of this call is still in flight when the worker is parked
  park_worker(wp);
  return rc;
}

}  // extern "C"
