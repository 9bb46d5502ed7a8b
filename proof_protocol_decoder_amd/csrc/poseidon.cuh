// poseidon.cuh -- Poseidon-Goldilocks permutation (width 12, x^7, 4+22+4 rounds) as device code.
//
// Restates upstream plonky2::hash::poseidon (reached from plonky_block_proof_gen/src/
// proof_gen.rs:44-52 via MerkleTree::new and the challenger); parameters in DESIGN.md section 3,
// constants from tools/gen_poseidon_constants.py (KAT-checked).  One lane owns one whole state
// (24 VGPRs): no cross-lane traffic, all 64 lanes busy in partial rounds too.  The round constants
// are wave-uniform, so they come through the scalar cache (s_load) and cost no VALU slots.
// Integer-ALU bound: no HBM-roofline claim is made for these kernels.
#pragma once
#include "gl.hpp"

namespace poseidon {

static __constant__ uint64_t RC[360] = {
#include "poseidon_rc.inc"
};

// x^7, any -> reduced
__device__ __forceinline__ uint64_t sbox(uint64_t x) {
  uint64_t x2 = gl::sqr(x);
  uint64_t x4 = gl::sqr(x2);
  uint64_t x3 = gl::mul(x2, x);
  return gl::mul(x3, x4);
}

// MDS = circulant(17,15,41,16,2,28,13,13,39,18,34,20) + diag(8,0,...).  Entries are < 2^6, so the
// 32-bit halves of the state are accumulated separately in 64 bits (no overflow: 12*41*2^32) and
// recombined with one small reduction per row.
__device__ __forceinline__ void mds(uint64_t (&s)[12]) {
  constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  uint32_t lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    lo[i] = (uint32_t)s[i];
    hi[i] = (uint32_t)(s[i] >> 32);
  }
#pragma unroll
  for (int r = 0; r < 12; r++) {
    uint64_t L = 0, H = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
      L += (uint64_t)lo[(i + r) % 12] * C[i];
      H += (uint64_t)hi[(i + r) % 12] * C[i];
    }
    if (r == 0) {
      L += (uint64_t)lo[0] * 8;
      H += (uint64_t)hi[0] * 8;
    }
    // value = L + H*2^32 < 2^74:  lo64 = L + (H<<32), top = (H>>32) + carry  (top < 2^11)
    uint64_t hs = H << 32;
    uint64_t lo64 = L + hs;
    uint64_t top = (H >> 32) + (lo64 < hs ? 1 : 0);
    uint64_t t1 = (top << 32) - top;  // top * (2^32-1), canonical
    uint64_t res = lo64 + t1;
    s[r] = res < t1 ? res + gl::EPS : res;
  }
}

__device__ __forceinline__ void permute(uint64_t (&s)[12]) {
  int rnd = 0;
#pragma unroll 1
  for (int k = 0; k < 4; k++, rnd++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = sbox(gl::add(s[i], RC[rnd * 12 + i]));
    mds(s);
  }
#pragma unroll 1
  for (int k = 0; k < 22; k++, rnd++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::add(s[i], RC[rnd * 12 + i]);
    s[0] = sbox(s[0]);
    mds(s);
  }
#pragma unroll 1
  for (int k = 0; k < 4; k++, rnd++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = sbox(gl::add(s[i], RC[rnd * 12 + i]));
    mds(s);
  }
}

// PoseidonHash::two_to_one
__device__ __forceinline__ void two_to_one(const uint64_t* l, const uint64_t* r, uint64_t* out) {
  uint64_t s[12];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    s[i] = l[i];
    s[4 + i] = r[i];
    s[8 + i] = 0;
  }
  permute(s);
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = gl::canon(s[i]);
}

}  // namespace poseidon
