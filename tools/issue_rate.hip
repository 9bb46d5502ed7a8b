// issue_rate.hip -- THROUGHPUT (not latency) of the VALU instructions the field kernels are made of:
// 24 independent accumulators per lane, 8 waves per SIMD, so the issue port is the only limit.
// Prints cycles per wave64 instruction per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/issue_rate tools/issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int ACC = 24, ITERS = 2048;

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a[ACC], b = seed + threadIdx.x, c = seed * 7 + 3;
  uint64_t w[ACC];
#pragma unroll
  for (int i = 0; i < ACC; i++) { a[i] = seed + i * 977 + threadIdx.x; w[i] = a[i]; }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ACC; i++) {
      if (OP == 0) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 1) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 2) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 4) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c) : "vcc");
      if (OP == 5) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(w[i]) : "v"(w[(i + 1) % ACC]));
      if (OP == 6) asm volatile("v_add_co_u32_e64 %0, s[10:11], %0, %1" : "+v"(a[i]) : "v"(b) : "s10", "s11");
      if (OP == 7) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(b));
      if (OP == 8) asm volatile("v_lshlrev_b32_e32 %0, 3, %0" : "+v"(a[i]));
      if (OP == 9) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 10) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 11) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[i]) : "v"(b));
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < ACC; i++) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
  const int cus = p.multiProcessorCount, blocks = cus * 8;  // 8 blocks x 4 waves = 8 waves per SIMD
  uint32_t* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  const char* names[] = {"v_add_u32", "v_mad_u32_u24", "v_dot4_u32_u8", "v_perm_b32", "v_mad_u64_u32", "v_lshl_add_u64",
                         "v_add_co_u32 (VOP3B, SGPR carry)", "v_cndmask_b32 (SGPR mask)", "v_lshlrev_b32", "v_add3_u32",
                         "v_mul_lo_u32", "v_mov_b32"};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(OP)                                                                                      \
  {                                                                                                  \
    k<OP><<<blocks, 256>>>(out, 1);                                                                  \
    hipDeviceSynchronize();                                                                          \
    float best = 1e30f;                                                                              \
    for (int r = 0; r < 3; r++) {                                                                    \
      hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, r + 2); hipEventRecord(e1); hipEventSynchronize(e1); \
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;                          \
    }                                                                                                \
    double wave_instrs_per_simd = (double)blocks * 4 / (cus * 4) * ACC * ITERS;                     \
    printf("%-34s %7.3f ms  %.2f cycles per wave64 instruction per SIMD (at 2.4 GHz)\n", names[OP], best, \
           best * 1e-3 * 2.4e9 / wave_instrs_per_simd);                                              \
  }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
  return 0;
}
