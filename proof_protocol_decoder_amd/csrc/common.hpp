// common.hpp -- error plumbing shared by the C-ABI translation units of libbpg.so.
#pragma once
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

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline unsigned ceil_div(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace bpg
