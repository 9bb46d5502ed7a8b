// mx_arith.cuh -- the two pieces of arithmetic every matrix-core kernel here ends with: the byte-plane sums an int8
// MFMA leaves in its i32 accumulators are recombined with shifts into two 64-bit halves (planes), and a pair of
// halves L + H * 2^32 is reduced to one lazily reduced Goldilocks word (reduce_rows, also the row reduction of the
// one-lane Poseidon MDS).  Used by poseidon.cuh, poseidon_mx.cuh (MDS layer) and ntt_mx.cuh (16-point DFT passes).
#pragma once
#include "gl.hpp"

namespace mxa {

typedef int v4i __attribute__((ext_vector_type(4)));

// N accumulator pairs (L and H below 2^48, so L + H*2^32 < 2^80) -> reduced words, carry-chain form in 4 instructions:
// H = h0 + h1*2^32 with h1 < 2^16, so L + H*2^32 = (L + h1*EPS) + h0*2^32 (mod p).  X = L + h1*EPS < 2^49 needs
// no carry; adding h0 to X's high word can carry once (c, weight 2^64 = EPS = 2^32 - 1): the result is
// (lo - c) + (hi + c)*2^32, where lo - c borrows (b) only if lo = 0 and then gives the 2^32 back: hi + (c & ~b).
// hi is below 2^17 whenever c is set (it wrapped), so nothing carries further.  The mask c & ~b is a scalar instruction.
template <int N>
__device__ __forceinline__ void reduce_rows(const uint64_t (&L)[N], const uint64_t (&H)[N], uint64_t (&out)[N]) {
  uint32_t l0[N], l1[N], h0[N], h1[N];
  uint64_t T[N];
  gl::cc::mask c1[N], b[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    T[i] = L[i]; h0[i] = (uint32_t)H[i]; h1[i] = (uint32_t)(H[i] >> 32);
  }
  gl::cc::mad_eps_cv(T, h1);      // X = L + h1*EPS (no carry: both below 2^42)
#pragma unroll
  for (int i = 0; i < N; i++) { l0[i] = (uint32_t)T[i]; l1[i] = (uint32_t)(T[i] >> 32); }
  gl::cc::add_co(l1, c1, h0);
  gl::cc::subb0_co(l0, b, c1);
#pragma unroll
  for (int i = 0; i < N; i++) c1[i] &= ~b[i];
  gl::cc::addc0_cv(l1, c1);
#pragma unroll
  for (int i = 0; i < N; i++) out[i] = gl::cc::mk64(l0[i], l1[i]);
}


// four non-negative plane sums (each < 2^23) -> a0 + a1*2^8 + a2*2^16 + a3*2^24 < 2^48: two shift-adds and one
// multiply-add (the compiler's own rendering of the 64-bit shift and add is five to six instructions)
__device__ __forceinline__ uint64_t planes(const v4i& d) {
  const uint32_t e = (uint32_t)d[0] + ((uint32_t)d[1] << 8), f = (uint32_t)d[2] + ((uint32_t)d[3] << 8);
  uint64_t r;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(f), "s"(65536u), "v"((uint64_t)e) : "vcc");
  return r;
}

}  // namespace mxa
