// mfma_probe.hip -- facts the matrix-core form of the Poseidon MDS layer rests on, checked on the device:
//  (1) operand / result lane maps of v_mfma_i32_16x16x64_i8 and v_mfma_i32_16x16x32_i8 with exact integer data;
//  (2) what v_permlane16_swap / v_permlane32_swap exchange;
//  (3) whether MFMAs issued between VALU instructions cost VALU issue slots (mixed-loop throughput).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_probe tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- (1) lane maps
// a_frag[l*16 + j], b_frag[l*16 + j]: byte j of lane l's 16-byte fragment.  d[l*4 + r]: result register r of lane l.
__global__ void mfma64_kernel(const int8_t* a_frag, const int8_t* b_frag, int* d) {
  const int l = threadIdx.x;
  v4i a, b, c = {0, 0, 0, 0};
  for (int w = 0; w < 4; w++) {
    a[w] = ((const int*)a_frag)[l * 4 + w];
    b[w] = ((const int*)b_frag)[l * 4 + w];
  }
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) d[l * 4 + r] = c[r];
}
__global__ void mfma32_kernel(const int8_t* a_frag, const int8_t* b_frag, int* d) {
  const int l = threadIdx.x;
  v4i c = {0, 0, 0, 0};
  const long a = ((const long*)a_frag)[l], b = ((const long*)b_frag)[l];
  c = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) d[l * 4 + r] = c[r];
}

// ---------------------------------------------------------------- (2) permlane swaps
__global__ void permlane_kernel(int* out) {
  const int l = threadIdx.x;
  int a = l, b = 100 + l;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  out[l] = a;
  out[64 + l] = b;
  a = l; b = 100 + l;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  out[128 + l] = a;
  out[192 + l] = b;
}

// ---------------------------------------------------------------- (3) mixed issue
constexpr int ACC = 24, ITERS = 1024;
// per iteration: 24 independent v_mad_u64_u32 and NM MFMAs (K = 64 form) on NM independent accumulators
template <int NM, int WAVES_PER_EU>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU)))
mix_kernel(uint32_t* out, uint32_t seed) {
  uint32_t b = seed + threadIdx.x, c = seed * 7 + 3;
  uint64_t w[ACC];
  v4i acc[NM > 0 ? NM : 1], fa, fb;
#pragma unroll
  for (int i = 0; i < ACC; i++) w[i] = seed + i * 977 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; i++) { fa[i] = seed * 3 + i + threadIdx.x; fb[i] = seed * 5 + i * threadIdx.x; }
#pragma unroll
  for (int i = 0; i < (NM > 0 ? NM : 1); i++) acc[i] = (v4i){0, 0, 0, 0};
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ACC; i++) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c) : "vcc");
      if (NM > 0 && (i % (ACC / (NM > 0 ? NM : 1))) == 0) {
        const int m = i / (ACC / (NM > 0 ? NM : 1));
        if (m < NM) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(fa), "v"(fb));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < ACC; i++) r ^= (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
#pragma unroll
  for (int i = 0; i < (NM > 0 ? NM : 1); i++) r ^= acc[i][0] ^ acc[i][1] ^ acc[i][2] ^ acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// the same loop with other MFMA shapes: KIND 1 = v_mfma_i32_16x16x32_i8 (8-byte operands), 2 = v_mfma_i32_32x32x32_i8
typedef int v16i __attribute__((ext_vector_type(16)));
template <int NM, int WAVES_PER_EU, int KIND>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU)))
mix2_kernel(uint32_t* out, uint32_t seed) {
  uint32_t b = seed + threadIdx.x, c = seed * 7 + 3;
  uint64_t w[ACC];
  v4i acc[NM], fa, fb;
  v16i big[KIND == 2 ? 2 : 1];
  long a8 = seed * 3 + threadIdx.x, b8 = seed * 5 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < ACC; i++) w[i] = seed + i * 977 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; i++) { fa[i] = seed * 3 + i + threadIdx.x; fb[i] = seed * 5 + i * threadIdx.x; }
#pragma unroll
  for (int i = 0; i < NM; i++) acc[i] = (v4i){0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; i++) { big[0][i] = 0; if (KIND == 2) big[KIND == 2 ? 1 : 0][i] = 0; }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < ACC; i++) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(b), "v"(c) : "vcc");
      if ((i % (ACC / NM)) == 0) {
        const int m = i / (ACC / NM);
        if (m < NM) {
          if (KIND == 1) asm volatile("v_mfma_i32_16x16x32_i8 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a8), "v"(b8));
          if (KIND == 2) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(big[m & 1]) : "v"(fa), "v"(fb));
        }
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < ACC; i++) r ^= (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
#pragma unroll
  for (int i = 0; i < NM; i++) r ^= acc[i][0] ^ acc[i][1] ^ acc[i][2] ^ acc[i][3];
#pragma unroll
  for (int i = 0; i < 16; i++) r ^= big[0][i] ^ big[KIND == 2 ? 1 : 0][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NM, int W, int KIND>
static void run_mix2(int cus, uint32_t* out) {
  const int blocks = cus * W;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mix2_kernel<NM, W, KIND><<<blocks, 256>>>(out, 1);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    hipEventRecord(e0); mix2_kernel<NM, W, KIND><<<blocks, 256>>>(out, r + 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("mix: %d waves/SIMD, 24 v_mad_u64_u32 + %d %s per iteration: %7.3f ms, %.1f cycles per iteration per SIMD (at 2.4 GHz)\n",
         W, NM, KIND == 1 ? "mfma_16x16x32_i8" : "mfma_32x32x32_i8", best, best * 1e-3 * 2.4e9 / ((double)W * ITERS));
}

template <int NM, int W>
static void run_mix(int cus, uint32_t* out) {
  const int blocks = cus * W;  // W blocks x 4 waves per CU = W waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mix_kernel<NM, W><<<blocks, 256>>>(out, 1);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    hipEventRecord(e0); mix_kernel<NM, W><<<blocks, 256>>>(out, r + 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double iters_per_simd = (double)W * ITERS;
  printf("mix: %d waves/SIMD, 24 v_mad_u64_u32 + %d mfma_16x16x64_i8 per iteration: %7.3f ms, %.1f cycles per iteration per SIMD (at 2.4 GHz)\n",
         W, NM, best, best * 1e-3 * 2.4e9 / iters_per_simd);
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
  printf("device %s, %d CUs\n", p.name, p.multiProcessorCount);
  // (1)
  {
    std::vector<int8_t> fa(64 * 16), fb(64 * 16);
    srand(12345);
    for (auto& v : fa) v = (int8_t)(rand() % 255 - 127);
    for (auto& v : fb) v = (int8_t)(rand() % 255 - 127);
    int8_t *da, *db; int* dd;
    hipMalloc(&da, fa.size()); hipMalloc(&db, fb.size()); hipMalloc(&dd, 64 * 4 * 4);
    hipMemcpy(da, fa.data(), fa.size(), hipMemcpyHostToDevice);
    hipMemcpy(db, fb.data(), fb.size(), hipMemcpyHostToDevice);
    std::vector<int> d(256);
    // K = 64: hypothesis A[row l&15][(l>>4, j)], B[(l>>4, j)][col l&15], pairing by (l>>4, j); D col = l&15, row = 4(l>>4)+r
    mfma64_kernel<<<1, 64>>>(da, db, dd);
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
      for (int r = 0; r < 4; r++) {
        const int col = l & 15, row = 4 * (l >> 4) + r;
        int want = 0;
        for (int kb = 0; kb < 4; kb++)
          for (int j = 0; j < 16; j++) want += (int)fa[(row + 16 * kb) * 16 + j] * (int)fb[(col + 16 * kb) * 16 + j];
        if (want != d[l * 4 + r]) bad++;
      }
    printf("mfma_i32_16x16x64_i8 lane map (A row = l&15, B col = l&15, k paired by (l>>4, byte j); D col = l&15, row = 4(l>>4)+reg): %s (%d of 256 differ)\n",
           bad ? "MISMATCH" : "ok", bad);
    mfma32_kernel<<<1, 64>>>(da, db, dd);  // fragment = first 8 bytes per lane: frag index l*8 + j
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    bad = 0;
    for (int l = 0; l < 64; l++)
      for (int r = 0; r < 4; r++) {
        const int col = l & 15, row = 4 * (l >> 4) + r;
        int want = 0;
        for (int kb = 0; kb < 4; kb++)
          for (int j = 0; j < 8; j++) want += (int)fa[(row + 16 * kb) * 8 + j] * (int)fb[(col + 16 * kb) * 8 + j];
        if (want != d[l * 4 + r]) bad++;
      }
    printf("mfma_i32_16x16x32_i8 lane map (same, 8 bytes per lane): %s (%d of 256 differ)\n", bad ? "MISMATCH" : "ok", bad);
  }
  // (2)
  {
    int* dd; hipMalloc(&dd, 256 * 4);
    permlane_kernel<<<1, 64>>>(dd);
    std::vector<int> o(256);
    hipMemcpy(o.data(), dd, 1024, hipMemcpyDeviceToHost);
    const char* nm[4] = {"permlane16_swap vdst", "permlane16_swap src ", "permlane32_swap vdst", "permlane32_swap src "};
    for (int k = 0; k < 4; k++) {
      printf("%s (in: vdst[l] = l, src[l] = 100 + l): rows of 16 lanes start with", nm[k]);
      for (int row = 0; row < 4; row++) printf(" %d", o[k * 64 + row * 16]);
      int contiguous = 1;
      for (int l = 0; l < 64; l++) if (o[k * 64 + l] != o[k * 64 + (l & ~15)] + (l & 15)) contiguous = 0;
      printf(" (%s)\n", contiguous ? "whole rows move" : "NOT row-wise");
    }
  }
  // (3)
  {
    uint32_t* out; hipMalloc(&out, (size_t)p.multiProcessorCount * 8 * 256 * 4);
    const int cus = p.multiProcessorCount;
    run_mix<0, 8>(cus, out); run_mix<2, 8>(cus, out); run_mix<4, 8>(cus, out); run_mix<6, 8>(cus, out); run_mix<8, 8>(cus, out);
    run_mix<0, 2>(cus, out); run_mix<4, 2>(cus, out); run_mix<8, 2>(cus, out);
    run_mix<0, 4>(cus, out); run_mix<4, 4>(cus, out); run_mix<8, 4>(cus, out);
    run_mix<0, 3>(cus, out); run_mix<8, 3>(cus, out);
    run_mix2<8, 2, 1>(cus, out); run_mix2<8, 4, 1>(cus, out); run_mix2<4, 2, 2>(cus, out); run_mix2<8, 2, 2>(cus, out);
    run_mix<0, 1>(cus, out); run_mix<4, 1>(cus, out);
  }
  return 0;
}
