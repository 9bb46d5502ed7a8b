// capi.cpp -- error state and process-level entry points of the C ABI (include/bpg.h).
// Error convention mirrors the reference's ProofGenError(String)
// (plonky_block_proof_gen/src/proof_gen.rs:16-36): a status code plus a thread-local message.
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

}  // namespace bpg

extern "C" {

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
