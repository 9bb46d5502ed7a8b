// common.hpp -- error plumbing shared by the C-ABI translation units of libbpg.so.
#pragma once
#include <exception>
#include <hip/hip_runtime.h>
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

struct KernelTimer {
  KernelTimer(int family, hipStream_t st, double alg_bytes);
  ~KernelTimer();
  void stop();  // record the end now (the destructor then does nothing more)
  int family;
  hipStream_t st;
  double bytes;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline unsigned ceil_div(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace bpg
