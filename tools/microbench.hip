// microbench.hip -- integer-ALU and HBM rates on gfx950 that size the field kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../proof_protocol_decoder_amd/csrc/gl.hpp"
#include "../proof_protocol_decoder_amd/csrc/poseidon.cuh"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

template <int OP>
__global__ void __launch_bounds__(256) alu_kernel(uint64_t* out, uint64_t seed) {
  uint64_t a[4], b = seed * 0x9E3779B97F4A7C15ULL + threadIdx.x;
  for (int k = 0; k < 4; k++) a[k] = seed + threadIdx.x * 4 + k + blockIdx.x;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (OP == 0) a[k] = (uint64_t)(uint32_t)a[k] * (uint32_t)b + a[k];            // v_mad_u64_u32
      if (OP == 1) a[k] = (uint32_t)a[k] * (uint32_t)b + (a[k] >> 32);               // v_mul_lo_u32 (+add)
      if (OP == 2) a[k] = __umulhi((uint32_t)a[k], (uint32_t)b) + (a[k] << 1);      // v_mul_hi_u32
      if (OP == 3) a[k] = a[k] + b;                                                  // 64-bit add
      if (OP == 4) a[k] = gl::mul(a[k], b);                                          // modmul
      if (OP == 5) a[k] = gl::addc(a[k], b & 0x7FFFFFFFFFFFFFFFULL);                 // canonical add
      if (OP == 6) a[k] = (a[k] << 3) + b;                                           // v_lshl_add_u64
      if (OP == 7) a[k] = __builtin_amdgcn_udot4((uint32_t)a[k], (uint32_t)b, (uint32_t)(a[k] >> 32), false);
      if (OP == 8) a[k] = __umul24((uint32_t)a[k], (uint32_t)b) + (uint32_t)(a[k] >> 32); // v_mad_u32_u24
      if (OP == 9) a[k] = (uint32_t)((uint32_t)a[k] + (uint32_t)b) ^ 0x5bd1e995u;             // v_add_u32 + v_xor_b32 (2 ops)
      if (OP == 10) {                                                                         // compiler-only modmul
        uint64_t lo, hi;
        gl::mul_wide(a[k], b, lo, hi);
        a[k] = gl::reduce128(lo, hi);
      }
    }
    if (OP == 12) {  // 64-bit add as two VOP3B carry ops with SGPR-pair carries (4 interleaved)
      uint32_t lo[4], hi[4], bl[4];
      gl::cc::mask c[4], cx[4];
#pragma unroll
      for (int k = 0; k < 4; k++) { lo[k] = (uint32_t)a[k]; hi[k] = (uint32_t)(a[k] >> 32); bl[k] = (uint32_t)b; }
      gl::cc::add_co(lo, c, bl);     // in place: lo += bl
      gl::cc::addc0_co(hi, cx, c);   // hi += carry
#pragma unroll
      for (int k = 0; k < 4; k++) a[k] = gl::cc::mk64(lo[k], hi[k]);
    }
    if (OP == 13) {  // v_cndmask_b32 x2 + v_cmp (compiler): select on a 64-bit compare
#pragma unroll
      for (int k = 0; k < 4; k++) a[k] = a[k] < b ? a[k] ^ 0x9E3779B97F4A7C15ULL : a[k] + 1;
    }
    if (OP == 11) {  // four interleaved carry-chain multiplies
      const uint64_t bb[4] = {b, b, b, b};
      uint64_t r[4];
      gl::mul_n<4>(a, bb, r);
#pragma unroll
      for (int k = 0; k < 4; k++) a[k] = r[k];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a[0] ^ a[1] ^ a[2] ^ a[3];
}

__global__ void __launch_bounds__(256) poseidon_kernel(uint64_t* out, uint64_t seed, int reps) {
  uint64_t s[12];
  for (int k = 0; k < 12; k++) s[k] = seed + threadIdx.x * 12 + k + blockIdx.x * 977;
  for (int r = 0; r < reps; r++) poseidon::permute(s);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0];
}

__global__ void copy_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}

template <typename F>
float time_ms(F f, int reps = 5) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, CUs %d, clock %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
  uint64_t* out;
  const int blocks = prop.multiProcessorCount * 8, threads = 256;
  CK(hipMalloc(&out, (size_t)prop.multiProcessorCount * 16 * threads * 8));  // largest launch below: 16 blocks/CU
  const char* names[] = {"v_mad_u64_u32", "v_mul_lo_u32+add", "v_mul_hi_u32+shl_add", "add_u64", "gl::mul (asm, 17 VALU+nop)",
                         "gl::addc", "v_lshl_add_u64", "v_dot4_u32_u8", "v_mad_u32_u24", "v_add_u32+v_xor_b32",
                         "modmul (compiler, 25 VALU)", "gl::mul_n<4> (17 VALU)", "add_co+addc_co (VOP3B, 2 ops)",
                         "cmp_u64+2 xor+2 cndmask+add64"};
  double ops = (double)blocks * threads * ITERS * 4;
  float ms;
#define RUN(OP) ms = time_ms([&] { alu_kernel<OP><<<blocks, threads>>>(out, 12345); }); \
  printf("%-28s %8.3f ms  %8.2f Gop/s  (%.1f lane-cycles/op at 2.4GHz x 256CU x 128 lanes)\n", names[OP], ms, ops / ms / 1e6, \
         (2.4e9 * prop.multiProcessorCount * 128) / (ops / ms * 1e3));
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13)
  for (int occ_blocks : {4, 8, 16}) {
    int pb = prop.multiProcessorCount * occ_blocks, reps = 64;
    ms = time_ms([&] { poseidon_kernel<<<pb, 256>>>(out, 99, reps); });
    printf("poseidon permute: blocks/CU=%d  %8.3f ms  %.3f Gperm/s\n", occ_blocks, ms, (double)pb * 256 * reps / ms / 1e6);
  }
  size_t bytes = (size_t)2 << 30;
  uint4 *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes));
  ms = time_ms([&] { copy_kernel<<<prop.multiProcessorCount * 16, 256>>>(a, b, bytes / 16); });
  printf("copy 2 GiB: %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * bytes / ms / 1e9);
  CK(hipDeviceSynchronize());
  return 0;
}
