// issue_rate6.hip -- round 2 extension of issue_rate.hip: which VALU instructions belong to the "cheap" class
// (v_add_u32 / v_mov_b32 measured 2.3-2.6 cycles per wave64 instruction in round 1) and which cost a full
// 4-cycle slot.  The answer decides the form of the Goldilocks carry chains (VCC-based VOP2 carry ops vs
// SGPR-pair VOP3B ones) and of the Poseidon MDS.
// 24 independent accumulators per lane, 8 waves per SIMD: only the issue port limits.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/issue_rate6 tools/issue_rate6.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int ACC = 24, ITERS = 2048;

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a[ACC], b = seed + threadIdx.x, c = seed * 7 + 3;
  uint64_t w[ACC];
#pragma unroll
  for (int i = 0; i < ACC; i++) { a[i] = seed + i * 977 + threadIdx.x; w[i] = a[i] * 0x100000001ull; }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ACC; i++) {
      if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c) : "vcc");
      if (OP == 1) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_u32_e32 %1, %2, %1" : "+v"(w[i]), "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
      if (OP == 2) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_u32_e32 %1, %2, %1\n\tv_add_u32_e32 %1, %3, %1" : "+v"(w[i]), "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
      if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_u32_e32 %1, %2, %1\n\tv_add_u32_e32 %1, %3, %1\n\tv_add_u32_e32 %1, %2, %1" : "+v"(w[i]), "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
      if (OP == 4) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mov_b32_e32 %1, %2" : "+v"(w[i]), "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
      if (OP == 5) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mov_b32_e32 %1, %2\n\tv_mov_b32_e32 %1, %3" : "+v"(w[i]), "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
      if (OP == 6) asm volatile("v_add_co_u32_e64 %0, s[10:11], %0, %2\n\tv_add_u32_e32 %1, %2, %1" : "+v"(a[i]), "+v"(a[(i+1)%ACC]) : "v"(b) : "s10", "s11");
      if (OP == 7) asm volatile("v_add_u32_e32 %0, %1, %0\n\tv_add_u32_e32 %0, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 8) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshrrev_b32_e32 %1, 3, %1\n\tv_and_b32_e32 %1, %2, %1" : "+v"(w[i]), "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
      if (OP == 9) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshl_add_u64 %1, %1, 1, %0" : "+v"(w[i]), "+v"(w[(i+1)%ACC]) : "v"(b), "v"(c) : "vcc");
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < ACC; i++) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static const char* names[] = {
    "mad only", "mad + 1 v_add_u32 (same wave, interleaved)", "mad + 2 v_add_u32", "mad + 3 v_add_u32", "mad + 1 v_mov_b32", "mad + 2 v_mov_b32", "add_co(sgpr) + 1 v_add_u32", "2 v_add_u32 only", "mad + v_lshrrev + v_and (2 cheap)", "mad + 1 v_lshl_add_u64 (expensive + expensive)"};

template <int OP>
static void run(uint32_t* out, int cus, int blocks, hipEvent_t e0, hipEvent_t e1) {
  k<OP><<<blocks, 256>>>(out, 1);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, r + 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double wave_ops_per_simd = (double)blocks * 4 / (cus * 4) * ACC * ITERS;
  printf("%-90s %7.3f ms  %.2f cycles per wave64 op per SIMD (at 2.4 GHz)\n", names[OP], best,
         best * 1e-3 * 2.4e9 / wave_ops_per_simd);
}

template <int OP>
static void run_all(uint32_t* out, int cus, int blocks, hipEvent_t e0, hipEvent_t e1) {
  run<OP>(out, cus, blocks, e0, e1);
  if constexpr (OP + 1 < 10) run_all<OP + 1>(out, cus, blocks, e0, e1);
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
  const int cus = p.multiProcessorCount, blocks = cus * 8;  // 8 blocks x 4 waves = 8 waves per SIMD
  uint32_t* out;
  if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("device clock %d kHz, %d CUs\n", p.clockRate, cus);
  run_all<0>(out, cus, blocks, e0, e1);
  return 0;
}
