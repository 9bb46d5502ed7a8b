// capi.cpp -- error state and process-level entry points of the C ABI (include/bpg.h).
// Error convention mirrors the reference's ProofGenError(String)
// (plonky_block_proof_gen/src/proof_gen.rs:16-36): a status code plus a thread-local message.
#include <atomic>
#include <deque>
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
// HIP events around a launch, on the launch's own stream.  Events are pooled per host thread and
// retired as soon as they have completed: creating two events per launch and keeping tens of
// thousands outstanding until the read-out made the runtime crawl at 24 prover streams.
namespace {
struct Pending { hipEvent_t e0, e1; int family; double bytes; };
struct FamilyStats { uint64_t launches = 0; double ms = 0, bytes = 0; };
struct ThreadProf {
  std::mutex mu;
  std::deque<Pending> pending;
  std::vector<hipEvent_t> free_events;
  FamilyStats stats[PROF_FAMILIES];
  // retire finished pairs from the front; with `wait` every pair
  void retire(bool wait) {
    while (!pending.empty()) {
      Pending& p = pending.front();
      if (wait ? hipEventSynchronize(p.e1) != hipSuccess : hipEventQuery(p.e1) != hipSuccess) {
        if (!wait) return;
      } else {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
          stats[p.family].launches++;
          stats[p.family].ms += ms;
          stats[p.family].bytes += p.bytes;
        }
      }
      free_events.push_back(p.e0);
      free_events.push_back(p.e1);
      pending.pop_front();
    }
  }
  hipEvent_t get() {
    if (!free_events.empty()) {
      hipEvent_t e = free_events.back();
      free_events.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
  }
};
std::mutex g_prof_mu;
std::vector<ThreadProf*> g_threads;  // never shrinks: prover threads are pooled for the life of a state
std::atomic<bool> g_prof_on{false};
ThreadProf& thread_prof() {
  thread_local ThreadProf* tp = nullptr;
  if (!tp) {
    tp = new ThreadProf();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_threads.push_back(tp);
  }
  return *tp;
}
}  // namespace
bool profile_on() { return g_prof_on.load(std::memory_order_relaxed); }
KernelTimer::KernelTimer(int f, hipStream_t s, double b, bool att) : attached(att), family(f), st(s), bytes(b) {
  if (!profile_on()) return;
  ThreadProf& tp = thread_prof();
  std::lock_guard<std::mutex> lk(tp.mu);
  e0 = tp.get();
  e1 = tp.get();
  if (!e0 || !e1) { e0 = e1 = nullptr; return; }
  if (!attached) (void)hipEventRecord(e0, st);
}
KernelTimer::~KernelTimer() { stop(); }
void KernelTimer::stop() {
  if (!e0) return;
  if (!attached) (void)hipEventRecord(e1, st);
  ThreadProf& tp = thread_prof();
  std::lock_guard<std::mutex> lk(tp.mu);
  if (attached && !launched) {
    // the scope left before its launch (an early return): the recycled events still hold an EARLIER interval, which
    // retire() would add to this family -- hand them back unrecorded
    tp.free_events.push_back(e0);
    tp.free_events.push_back(e1);
    e0 = e1 = nullptr;
    return;
  }
  tp.pending.push_back(Pending{e0, e1, family, bytes});
  if (tp.pending.size() >= 64) tp.retire(false);
  e0 = e1 = nullptr;
}
// everything recorded so far, summed over threads; `reset` also clears the sums
static void profile_collect(FamilyStats (&out)[PROF_FAMILIES], bool reset) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (ThreadProf* tp : g_threads) {
    std::lock_guard<std::mutex> lk2(tp->mu);
    tp->retire(true);
    for (int f = 0; f < PROF_FAMILIES; f++) {
      out[f].launches += tp->stats[f].launches;
      out[f].ms += tp->stats[f].ms;
      out[f].bytes += tp->stats[f].bytes;
      if (reset) tp->stats[f] = FamilyStats();
    }
  }
}

}  // namespace bpg

extern "C" {

void bp_profile_enable(int on) { bpg::g_prof_on.store(on != 0); }
void bp_profile_reset(void) {
  bpg::FamilyStats scratch[bpg::PROF_FAMILIES];
  bpg::profile_collect(scratch, true);
}
int bp_profile_read(int family, uint64_t* launches, double* total_ms, double* total_alg_bytes) {
  if (family < 0 || family >= bpg::PROF_FAMILIES) return bpg::fail(BP_ERR_INVALID_INPUT, "unknown kernel family %d", family);
  bpg::FamilyStats all[bpg::PROF_FAMILIES];
  bpg::profile_collect(all, false);
  if (launches) *launches = all[family].launches;
  if (total_ms) *total_ms = all[family].ms;
  if (total_alg_bytes) *total_alg_bytes = all[family].bytes;
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

// 0 undecided, 1 interrupt-driven host waits (hipDeviceScheduleBlockingSync), 2 left as the process had it
static std::atomic<int> g_wait_mode[64];
static std::mutex g_wait_mu;

int bp_use_blocking_sync(int device) {
  // On this ROCm every host wait (hipStreamSynchronize, hipEventSynchronize -- even on an event made
  // with hipEventBlockingSync) spins at 100 % of a core unless the device was given
  // hipDeviceScheduleBlockingSync (tools/wait_probe.hip: 100 % -> 1 %).  With one prover thread per
  // stream the spinning threads fill the box's cores and more streams than cores lose throughput.
  // Decided once per device, by the first caller -- an explicit call (bench.py, before torch creates the context) or
  // the first worker this library creates (Worker::init) -- and never changed afterwards.  The mode can only be
  // switched while the process has not used the device yet: switching it under queues that already carried work
  // (the application's null stream, a parked worker's stream) makes a later device-wide wait -- hipFree,
  // hipDeviceSynchronize -- never return (tools/hang_probe.py; tests/test_gpu_proofgen.py).  A device the process
  // has already used therefore keeps the mode it has; the library's own waits then poll and sleep
  // (Worker::wait, prover.cpp) instead of calling the runtime's spinning wait.
  if (device < 0 || device >= 64) return bpg::fail(BP_ERR_DEVICE, "device %d out of range", device);
  if (g_wait_mode[device].load(std::memory_order_acquire)) return BP_OK;
  // The whole decision is one critical section: a second thread must not create its stream (Worker::init) between
  // this thread's "decided" and its hipSetDeviceFlags -- that sequence is the round-3 hipFree hang.
  std::lock_guard<std::mutex> lk(g_wait_mu);
  if (g_wait_mode[device].load(std::memory_order_acquire)) return BP_OK;
  struct Decide {  // the mode is published last, on every way out
    std::atomic<int>& slot;
    int mode = 2;
    ~Decide() { slot.store(mode, std::memory_order_release); }
  } decide{g_wait_mode[device]};
  unsigned int flags = 0;
  int active = 0;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wdeprecated-declarations"
  // (deprecated as a CUDA driver-API equivalent; it is the one call that tells whether the process has used the device)
  hipError_t e = hipDevicePrimaryCtxGetState(device, &flags, &active);
#pragma clang diagnostic pop
  if (e != hipSuccess) return bpg::fail(BP_ERR_DEVICE, "hipDevicePrimaryCtxGetState(%d): %s", device, hipGetErrorString(e));
  if ((flags & hipDeviceScheduleMask) == hipDeviceScheduleBlockingSync) {
    decide.mode = 1;
    return BP_OK;
  }
  if (active) return BP_OK;  // in use already: not ours to switch
  int prev = 0;
  (void)hipGetDevice(&prev);
  e = hipSetDevice(device);
  if (e == hipSuccess) e = hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
  (void)hipSetDevice(prev);
  if (e != hipSuccess) return bpg::fail(BP_ERR_DEVICE, "hipSetDeviceFlags(blocking sync) on device %d: %s", device, hipGetErrorString(e));
  decide.mode = 1;
  return BP_OK;
}

int bp_host_wait_mode(int device) { return device >= 0 && device < 64 ? g_wait_mode[device].load() : 0; }

}  // extern "C"
