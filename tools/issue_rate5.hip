// issue_rate5.hip -- round 2 extension of issue_rate.hip: which VALU instructions belong to the "cheap" class
// (v_add_u32 / v_mov_b32 measured 2.3-2.6 cycles per wave64 instruction in round 1) and which cost a full
// 4-cycle slot.  The answer decides the form of the Goldilocks carry chains (VCC-based VOP2 carry ops vs
// SGPR-pair VOP3B ones) and of the Poseidon MDS.
// 24 independent accumulators per lane, 8 waves per SIMD: only the issue port limits.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/issue_rate5 tools/issue_rate5.hip
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
      if (OP == 0) asm volatile("v_cmp_lt_u64_e32 vcc, %2, %3\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_cndmask_b32_e32 %1, %1, %4, vcc" : "+v"(a[i]), "+v"(a[(i+1)%ACC]) : "v"(w[i]), "v"(w[(i+1)%ACC]), "v"(b) : "vcc");
      if (OP == 1) asm volatile("v_cmp_lt_u64_e32 vcc, %2, %3\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_cndmask_b32_e32 %1, %1, %4, vcc\n\tv_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_cndmask_b32_e32 %1, %1, %4, vcc" : "+v"(a[i]), "+v"(a[(i+1)%ACC]) : "v"(w[i]), "v"(w[(i+1)%ACC]), "v"(b) : "vcc");
      if (OP == 2) asm volatile("v_cmp_lt_u64_e32 vcc, %2, %3\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_add_u32_e32 %1, %4, %1\n\tv_cndmask_b32_e32 %1, %1, %4, vcc" : "+v"(a[i]), "+v"(a[(i+1)%ACC]) : "v"(w[i]), "v"(w[(i+1)%ACC]), "v"(b) : "vcc");
      if (OP == 3) asm volatile("v_cmp_lt_u64_e64 s[10:11], %2, %3\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %4, s[10:11]\n\tv_cndmask_b32_e64 %1, %1, %4, s[10:11]" : "+v"(a[i]), "+v"(a[(i+1)%ACC]) : "v"(w[i]), "v"(w[(i+1)%ACC]), "v"(b) : "s10", "s11");
      if (OP == 4) asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n\tv_cmp_lt_u64_e32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 %2, %2, %3, vcc\n\tv_cndmask_b32_e32 %2, %2, %3, vcc" : "+v"(w[i]), "+v"(w[(i+1)%ACC]), "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 5) asm volatile("v_add_co_u32_e64 %0, s[10:11], %0, %2\n\ts_nop 1\n\tv_addc_co_u32_e64 %1, s[12:13], %1, %2, s[10:11]\n\ts_nop 1\n\tv_cndmask_b32_e64 %3, 0, -1, s[12:13]\n\tv_add_co_u32_e64 %0, s[10:11], %0, %3\n\ts_nop 1\n\tv_addc_co_u32_e64 %1, s[12:13], %1, 0, s[10:11]" : "+v"(a[i]), "+v"(a[(i+1)%ACC]) : "v"(b), "v"(c) : "s10", "s11", "s12", "s13");
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < ACC; i++) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static const char* names[] = {
    "A: cmp_lt_u64 vcc; nop1; 2x cndmask_e32 (3 ops)", "B: cmp_lt_u64 vcc; nop1; 4x cndmask_e32 (5 ops)", "C: cmp vcc; nop1; cndmask_e32; v_add_u32; cndmask_e32 (4 ops)", "D: cmp_e64 s[10:11]; nop1; 2x cndmask_e64 (3 ops)", "E: lshl_add_u64; cmp vcc; nop1; 2x cndmask_e32 (4 ops)", "F: carry-chain add: add_co,addc_co,sel_eps,add_co,addc0 sgpr (5 ops, nop1 x3)"};

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
  if constexpr (OP + 1 < 6) run_all<OP + 1>(out, cus, blocks, e0, e1);
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
