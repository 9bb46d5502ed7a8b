// issue_rate.hip -- THROUGHPUT (not latency) of the VALU instructions and instruction patterns the field kernels are
// made of: 24 independent accumulators per lane, 8 waves per SIMD, so the issue port is the only limit.  Prints cycles
// per wave64 instruction per SIMD.  One probe, six instruction sets (they were six files in round 2):
//   1  the basic classes (round 1: v_add_u32 / v_mov_b32 2.3-2.6, everything else 4.1-4.8)
//   2  which instructions belong to the cheap class: 36 forms (VOP2 / VOP3, VCC and SGPR-pair carries, DPP, packed,
//      float, 64-bit) -- decides the form of the carry chains and of the Poseidon MDS
//   3  v_cndmask on VCC against an SGPR mask, and what an s_nop after an instruction costs
//   4  v_cndmask mask sources (VCC never written, low / high SGPR pairs) and compare + select pairs
//   5  the compiler's 64-bit compare-and-select pattern (cmp vcc; 2-4 x cndmask_e32) against the SGPR-pair form:
//      back-to-back v_cndmask_b32_e32 on VCC stall (28 cycles against 13)
//   6  mixes of one full-rate instruction with 1-3 cheap ones in the same wave (do cheap instructions ride along?)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/issue_rate tools/issue_rate.hip;  run: tools/issue_rate [set | all]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace set1 {
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

int run() {
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
#undef RUN
}  // namespace set1

namespace set2 {
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
      if (OP == 0) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 1) asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 2) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 3) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 4) asm volatile("v_lshrrev_b32_e32 %0, 1, %0" : "+v"(a[i]));
      if (OP == 5) asm volatile("v_add_co_u32_e32 %0, vcc, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 6) asm volatile("v_addc_co_u32_e32 %0, vcc, %1, %0, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 7) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 8) asm volatile("v_add_u32_e64 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 9) asm volatile("v_add_u32_e32 %0, 0x12345, %0" : "+v"(a[i]));
      if (OP == 10) asm volatile("v_mul_u32_u24_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 11) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 12) asm volatile("v_mad_u64_u32 %0, vcc, %1, 17, %0" : "+v"(w[i]) : "v"(b) : "vcc");
      if (OP == 13) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c) : "s10", "s11");
      if (OP == 14) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 15) asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b));
      if (OP == 16) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (OP == 17) asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
      if (OP == 18) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 19) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 20) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 21) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(w[i]) : "v"(w[(i + 1) % ACC]));
      if (OP == 22) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(w[i]) : "v"(w[(i + 1) % ACC]));
      if (OP == 23) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(w[i]));
      if (OP == 24) asm volatile("v_sub_co_u32_e32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 25) asm volatile("v_subb_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 26) asm volatile("v_addc_co_u32_e64 %0, s[10:11], %0, %1, s[10:11]" : "+v"(a[i]) : "v"(b) : "s10", "s11");
      if (OP == 27) asm volatile("v_or_b32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 28) asm volatile("v_bfe_u32 %0, %0, 3, 22" : "+v"(a[i]));
      if (OP == 29) asm volatile("v_alignbit_b32 %0, %0, %1, 22" : "+v"(a[i]) : "v"(b));
      if (OP == 30) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 31) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c) : "vcc");
      // a 64-bit add as the compiler writes it (VCC chain), counted as ONE operation of two instructions
      if (OP == 32) asm volatile("v_add_co_u32_e32 %0, vcc, %2, %0\n\tv_addc_co_u32_e32 %1, vcc, %3, %1, vcc"
                                 : "+v"(a[i]), "+v"(a[(i + 1) % ACC]) : "v"(b), "v"(c) : "vcc");
      if (OP == 33) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(w[i]) : "v"(w[(i + 1) % ACC]));
      if (OP == 34) asm volatile("v_min_u32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 35) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < ACC; i++) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static const char* names[] = {
    "v_add_u32_e32", "v_sub_u32_e32", "v_and_b32_e32", "v_xor_b32_e32", "v_lshrrev_b32_e32",
    "v_add_co_u32_e32 (VCC)", "v_addc_co_u32_e32 (VCC)", "v_cndmask_b32_e32 (VCC)", "v_add_u32_e64",
    "v_add_u32_e32 literal", "v_mul_u32_u24_e32", "v_mul_hi_u32", "v_mad_u64_u32 inline-const", "v_mad_u64_u32 sgpr-carry",
    "v_lshl_add_u32", "v_add_lshl_u32", "v_mov_b32_dpp quad_perm", "v_add_u32_dpp quad_perm", "v_pk_add_u16",
    "v_add_f32_e32", "v_fma_f32", "v_pk_fma_f32", "v_fma_f64", "v_lshlrev_b64", "v_sub_co_u32_e32 (VCC)",
    "v_subb_co_u32_e32 (VCC)", "v_addc_co_u32_e64 (SGPR in+out)", "v_or_b32_e32", "v_bfe_u32", "v_alignbit_b32",
    "v_mul_lo_u32", "v_mad_u64_u32 (VCC)", "64-bit add = add_co+addc (2 instr)", "v_lshl_add_u64", "v_min_u32_e32",
    "v_mad_u32_u24"};

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
  if constexpr (OP + 1 < 36) run_all<OP + 1>(out, cus, blocks, e0, e1);
}

int run() {
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
#undef RUN
}  // namespace set2

namespace set3 {
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

int run() {
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
#undef RUN
}  // namespace set3

namespace set4 {
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
      if (OP == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
      if (OP == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(b));
      if (OP == 3) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[100:101]" : "+v"(a[i]) : "v"(b) : "s100", "s101");
      if (OP == 4) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 5) asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(b) : "s10", "s11");
      if (OP == 6) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 7) asm volatile("v_add_co_u32_e32 %0, vcc, %1, %0\n\tv_addc_co_u32_e32 %0, vcc, %1, %0, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 8) asm volatile("v_add_co_u32_e64 %0, s[10:11], %1, %0\n\ts_nop 1\n\tv_addc_co_u32_e64 %0, s[10:11], %1, %0, s[10:11]" : "+v"(a[i]) : "v"(b) : "s10", "s11");
      if (OP == 9) asm volatile("v_add_co_u32_e64 %0, s[10:11], %1, %0\n\tv_addc_co_u32_e64 %0, s[10:11], %1, %0, s[10:11]" : "+v"(a[i]) : "v"(b) : "s10", "s11");
      if (OP == 10) asm volatile("v_cndmask_b32_dpp %0, %0, %1, vcc quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
      if (OP == 11) asm volatile("v_readfirstlane_b32 s10, %0" : : "v"(a[i]) : "s10");
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < ACC; i++) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static const char* names[] = {
    "v_cndmask_b32_e32 vcc (never written)", "v_cndmask_b32_e64 vcc (never written)", "v_cndmask_b32_e64 s[10:11]", "v_cndmask_b32_e64 s[100:101]", "pair: v_cmp_lt_u32_e32 vcc ; s_nop 1 ; v_cndmask_e32 vcc", "pair: v_cmp_lt_u32_e64 s[10:11] ; s_nop 1 ; v_cndmask_e64 s[10:11]", "pair: v_cmp_lt_u32_e32 vcc ; s_nop 1 ; v_cndmask_e64 vcc", "pair: v_add_co_u32_e32 vcc ; v_addc_co_u32_e32 vcc", "pair: v_add_co_u32_e64 s[10:11] ; s_nop 1 ; v_addc_co_u32_e64 s[10:11]", "pair: v_add_co_u32_e64 s[10:11] ; v_addc_co_u32_e64 s[10:11] (no nop)", "v_cndmask_b32_dpp vcc", "v_readfirstlane_b32"};

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
  printf("%-70s %7.3f ms  %.2f cycles per wave64 op per SIMD (at 2.4 GHz)\n", names[OP], best,
         best * 1e-3 * 2.4e9 / wave_ops_per_simd);
}

template <int OP>
static void run_all(uint32_t* out, int cus, int blocks, hipEvent_t e0, hipEvent_t e1) {
  run<OP>(out, cus, blocks, e0, e1);
  if constexpr (OP + 1 < 12) run_all<OP + 1>(out, cus, blocks, e0, e1);
}

int run() {
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
#undef RUN
}  // namespace set4

namespace set5 {
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

int run() {
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
#undef RUN
}  // namespace set5

namespace set6 {
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

int run() {
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
#undef RUN
}  // namespace set6

int main(int argc, char** argv) {
  const bool all = argc > 1 && !strcmp(argv[1], "all");
  const int set = argc > 1 && !all ? atoi(argv[1]) : 1;
  int rc = 0;
  if (all || set == 1) rc |= set1::run();
  if (all || set == 2) rc |= set2::run();
  if (all || set == 3) rc |= set3::run();
  if (all || set == 4) rc |= set4::run();
  if (all || set == 5) rc |= set5::run();
  if (all || set == 6) rc |= set6::run();
  return rc;
}
