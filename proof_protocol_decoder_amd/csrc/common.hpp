// common.hpp -- error plumbing shared by the C-ABI translation units of libbpg.so.
#pragma once
#include <exception>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdarg>
#include <cstdio>
#include <string>
#include "../../include/bpg.h"

namespace bpg {

std::string& last_error_ref();
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define BPG_HIP(expr)                                                                        \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return bpg::fail(BP_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                       __FILE__, __LINE__);                                                  \
  } while (0)

#define BPG_LAUNCH_CHECK() BPG_HIP(hipGetLastError())

// Optional per-kernel-family HIP-event timing (bench.py's roofline leg).  Off by default: when off the
// guard is two branches.  Families: 0 = LDE coset NTT (DIT, LDS-resident), 1 = inverse NTT (DIF).
// family 2 counts permutations, not bytes; 3..6: the other HBM-class kernels of SURVEY.md section 8(d); 7 + air_id: K5
enum { PROF_LDE_DIT = 0, PROF_INTT_DIF = 1, PROF_LEAF_HASH = 2, PROF_FRI_FOLD = 3, PROF_OPENINGS = 4, PROF_FRI_COMBINE = 5,
       PROF_AUX = 6, PROF_K5 = 7 /* + air_id, nine AIRs */, PROF_FAMILIES = 16 };
bool profile_on();
// Function-try-block tail of every allocating extern "C" entry: nothing may unwind across the C ABI
// (include/bpg.h); an exception becomes BP_ERR_DEVICE with its message.
#define BPG_ABI_CATCH(name)                                                                          \
  catch (const std::exception& e) { return ::bpg::fail(BP_ERR_DEVICE, name ": %s", e.what()); }      \
  catch (...) { return ::bpg::fail(BP_ERR_DEVICE, name ": unknown exception"); }

// attached = false: the interval between two hipEventRecord calls around whatever is launched in its scope (the
// kernel plus the dispatch gap, ~3 us).  attached = true: the scope holds ONE launch made with BPG_LAUNCH_TIMED, which
// hands e0 / e1 to hipExtLaunchKernelGGL as the kernel's own start / stop events: the interval is the kernel's alone,
// which is what rocprofv3 reports for it.
struct KernelTimer {
  KernelTimer(int family, hipStream_t st, double alg_bytes, bool attached = false);
  ~KernelTimer();
  void stop();  // record the end now (the destructor then does nothing more)
  bool attached = false;
  bool launched = false;  // attached: BPG_LAUNCH_TIMED ran (a scope left early has recorded nothing into e0 / e1)
  int family;
  hipStream_t st;
  double bytes;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};

// one kernel launch inside an attached KernelTimer's scope (KERNEL with template arguments: HIP_KERNEL_NAME(k<a, b>))
#define BPG_LAUNCH_TIMED(kt, KERNEL, grid, block, lds, st, ...)                                                          \
  do {                                                                                                                  \
    if ((kt).e0) {                                                                                                      \
      (kt).launched = true;                                                                                             \
      hipExtLaunchKernelGGL(KERNEL, dim3(grid), dim3(block), (uint32_t)(lds), st, (kt).e0, (kt).e1, 0, __VA_ARGS__);    \
    }                                                                                                                   \
    else KERNEL<<<grid, block, lds, st>>>(__VA_ARGS__);                                                                 \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline unsigned ceil_div(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace bpg
