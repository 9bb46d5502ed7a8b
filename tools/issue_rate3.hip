// issue_rate3.hip -- round 2 extension of issue_rate.hip: which VALU instructions belong to the "cheap" class
// (v_add_u32 / v_mov_b32 measured 2.3-2.6 cycles per wave64 instruction in round 1) and which cost a full
// 4-cycle slot.  The answer decides the form of the Goldilocks carry chains (VCC-based VOP2 carry ops vs
// SGPR-pair VOP3B ones) and of the Poseidon MDS.
// 24 independent accumulators per lane, 8 waves per SIMD: only the issue port limits.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/issue_rate3 tools/issue_rate3.hip
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
      if (OP == 0) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
      if (OP == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(b));
      if (OP == 2) asm volatile("v_add_u32_e32 %0, %1, %0\n\ts_nop 0" : "+v"(a[i]) : "v"(b));
      if (OP == 3) asm volatile("v_add_u32_e32 %0, %1, %0\n\ts_nop 1" : "+v"(a[i]) : "v"(b));
      if (OP == 4) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\ts_nop 0" : "+v"(w[i]) : "v"(b), "v"(c) : "vcc");
      if (OP == 5) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\ts_nop 1" : "+v"(w[i]) : "v"(b), "v"(c) : "vcc");
      if (OP == 6) asm volatile("v_lshlrev_b32_e32 %0, 3, %0" : "+v"(a[i]));
      if (OP == 7) asm volatile("v_cmp_lt_u64_e32 vcc, %0, %1" : : "v"(w[i]), "v"(w[(i + 1) % ACC]) : "vcc");
      if (OP == 8) asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %1" : : "v"(a[i]), "v"(b) : "s10", "s11");
      if (OP == 9) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 10) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 11) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 12) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 13) asm volatile("v_subrev_u32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 14) asm volatile("v_ashrrev_i32_e32 %0, 1, %0" : "+v"(a[i]));
      if (OP == 15) asm volatile("v_max_u32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 16) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 17) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 18) asm volatile("v_cvt_f32_u32_e32 %0, %0" : "+v"(a[i]));
      if (OP == 19) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(a[i]) : "v"(b));
      if (OP == 20) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(a[i]) : "v"(b));
      if (OP == 21) asm volatile("v_mov_b64 %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) % ACC]));
      if (OP == 22) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 23) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mov_b32_e32 %3, 0" : "+v"(w[i]), "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < ACC; i++) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static const char* names[] = {
    "v_cndmask_b32_e32 (VCC, no clobber)", "v_cndmask_b32_e64 (SGPR mask)", "v_add_u32 ; s_nop 0", "v_add_u32 ; s_nop 1", "v_mad_u64_u32 ; s_nop 0", "v_mad_u64_u32 ; s_nop 1", "v_lshlrev_b32_e32", "v_cmp_lt_u64_e32 (VCC)", "v_cmp_lt_u32_e64 (SGPR)", "v_add3_u32", "v_lshl_or_b32", "v_and_or_b32", "v_bitop3_b32", "v_subrev_u32_e32", "v_ashrrev_i32_e32", "v_max_u32_e32", "v_mul_f32_e32", "v_fmac_f32_e32", "v_cvt_f32_u32_e32", "v_mov_b32 sdwa", "v_add_u32 sdwa", "v_mov_b64", "v_pk_mul_lo_u16", "v_mad_u64_u32 then v_mov (pair build)"};

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
  printf("%-36s %7.3f ms  %.2f cycles per wave64 op per SIMD (at 2.4 GHz)\n", names[OP], best,
         best * 1e-3 * 2.4e9 / wave_ops_per_simd);
}

template <int OP>
static void run_all(uint32_t* out, int cus, int blocks, hipEvent_t e0, hipEvent_t e1) {
  run<OP>(out, cus, blocks, e0, e1);
  if constexpr (OP + 1 < 24) run_all<OP + 1>(out, cus, blocks, e0, e1);
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
