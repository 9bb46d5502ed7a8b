// perm_latency.hip -- how long ONE Poseidon permutation takes on the device when nothing can be overlapped with it:
// a single wave running N dependent permutations (the Fiat-Shamir challenger is exactly that: a strictly sequential
// duplex sponge), in the three kernel forms of csrc/ (one lane per state, four lanes per state, matrix-core form with
// one set of 16 states).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I proof_protocol_decoder_amd/csrc
//   -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/perm_latency tools/perm_latency.hip     (__graft_entry__.build does it)
// Prints microseconds per permutation; profiles/r5_transcript_latency.txt holds a run next to the host's figure.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "gl.hpp"
#include "poseidon.cuh"
#include "poseidon_mx.cuh"

__global__ void __launch_bounds__(64) lane_chain(uint64_t* io, int n) {
  uint64_t s[12];
  for (int k = 0; k < 12; k++) s[k] = io[threadIdx.x * 12 + k];
  for (int i = 0; i < n; i++) poseidon::permute(s);
  for (int k = 0; k < 12; k++) io[threadIdx.x * 12 + k] = gl::canon(s[k]);
}
__global__ void __launch_bounds__(64) quad_chain(uint64_t* io, int n) {
  __shared__ uint64_t rc[360];
  for (uint32_t i = threadIdx.x; i < 360; i += blockDim.x) rc[i] = poseidon::RC[i];
  __syncthreads();
  const poseidon::QuadCtx qc = poseidon::quad_ctx();
  uint64_t e[3];
  for (int a = 0; a < 3; a++) e[a] = io[(threadIdx.x >> 2) * 12 + qc.q + 4 * a];
  for (int i = 0; i < n; i++) poseidon::permute_quad(e, qc, rc);
  for (int a = 0; a < 3; a++) io[(threadIdx.x >> 2) * 12 + qc.q + 4 * a] = gl::canon(e[a]);
}
__global__ void __launch_bounds__(64) mx_chain(uint64_t* io, int n) {
  __shared__ __attribute__((aligned(16))) uint32_t cin[poseidon::mx::CIN_WORDS];
  poseidon::mx::build_cin(cin);
  __syncthreads();
  const poseidon::mx::Ctx c = poseidon::mx::make_ctx(cin);
  uint64_t e[1][3];
  for (int a = 0; a < 3; a++) e[0][a] = io[(threadIdx.x & 15) * 12 + c.kb + 4 * a];
  for (int i = 0; i < n; i++) poseidon::mx::permute<1>(e, c);
  for (int a = 0; a < 3; a++) io[(threadIdx.x & 15) * 12 + c.kb + 4 * a] = gl::canon(e[0][a]);
}

template <class K>
static double time_it(K kernel, uint64_t* d, int n) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  kernel<<<1, 64>>>(d, 8);
  hipDeviceSynchronize();
  double best = 1e30;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(a);
    kernel<<<1, 64>>>(d, n);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best * 1e3 / n;
}
int main() {
  uint64_t* d = nullptr;
  hipMalloc(&d, 64 * 12 * 8);
  hipMemset(d, 1, 64 * 12 * 8);
  const int n = 2000;
  printf("one wave, %d dependent permutations, us per permutation:\n", n);
  printf("  one lane per state (poseidon::permute)          %.2f\n", time_it(lane_chain, d, n));
  printf("  four lanes per state (poseidon::permute_quad)   %.2f\n", time_it(quad_chain, d, n));
  printf("  matrix-core form, one set (mx::permute<1>)      %.2f\n", time_it(mx_chain, d, n));
  hipFree(d);
  return 0;
}
