// capi.cpp -- error state and process-level entry points of the C ABI (include/bpg.h).
// Error convention mirrors the reference's ProofGenError(String)
// (plonky_block_proof_gen/src/proof_gen.rs:16-36): a status code plus a thread-local message.
#include <atomic>
#include <mutex>
#include <vector>
#include "common.hpp"

namespace bpg {

std::string& last_error_ref() {
  static thread_local std::string s;
  return s;
}

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return code;
}

// ---- kernel-family timing ----
namespace {
struct Pending { hipEvent_t e0, e1; int family; double bytes; };
std::mutex g_prof_mu;
std::vector<Pending> g_pending;
std::atomic<bool> g_prof_on{false};
struct FamilyStats { uint64_t launches = 0; double ms = 0, bytes = 0; } g_stats[PROF_FAMILIES];
}  // namespace
bool profile_on() { return g_prof_on.load(std::memory_order_relaxed); }
KernelTimer::KernelTimer(int f, hipStream_t s, double b) : family(f), st(s), bytes(b) {
  if (!profile_on()) return;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
  (void)hipEventRecord(e0, st);
}
KernelTimer::~KernelTimer() {
  if (!e0) return;
  (void)hipEventRecord(e1, st);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_pending.push_back(Pending{e0, e1, family, bytes});
}
static void profile_drain() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& p : g_pending) {
    float ms = 0;
    if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      g_stats[p.family].launches++;
      g_stats[p.family].ms += ms;
      g_stats[p.family].bytes += p.bytes;
    }
    (void)hipEventDestroy(p.e0);
    (void)hipEventDestroy(p.e1);
  }
  g_pending.clear();
}

}  // namespace bpg

extern "C" {

void bp_profile_enable(int on) { bpg::g_prof_on.store(on != 0); }
void bp_profile_reset(void) {
  bpg::profile_drain();
  for (auto& s : bpg::g_stats) s = bpg::FamilyStats();
}
int bp_profile_read(int family, uint64_t* launches, double* total_ms, double* total_alg_bytes) {
  if (family < 0 || family >= bpg::PROF_FAMILIES) return bpg::fail(BP_ERR_INVALID_INPUT, "unknown kernel family %d", family);
  bpg::profile_drain();
  if (launches) *launches = bpg::g_stats[family].launches;
  if (total_ms) *total_ms = bpg::g_stats[family].ms;
  if (total_alg_bytes) *total_alg_bytes = bpg::g_stats[family].bytes;
  return BP_OK;
}

const char* bp_last_error(void) { return bpg::last_error_ref().c_str(); }
const char* bp_version(void) { return "bpg 0.1 (gfx950)"; }

int bp_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    if (e == hipErrorNoDevice) return 0;
    return bpg::fail(BP_ERR_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  }
  return n;
}

}  // extern "C"
