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
#include "mx_arith.cuh"

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

// x^7 on N (3 or 4) independent words with the interleaved multiply (gl::mul_n)
template <int N>
__device__ __forceinline__ void sbox_n(uint64_t (&x)[N]) {
  uint64_t x2[N], x4[N], x3[N];
  gl::mul_n<N>(x, x, x2);
  gl::mul_n<N>(x2, x2, x4);
  gl::mul_n<N>(x2, x, x3);
  gl::mul_n<N>(x3, x4, x);
}
// Twelve S-boxes as four groups of THREE, one group's carry masks at a time: next to the 24 round-constant SGPRs the
// masks of two overlapped groups of four do not fit the SGPR file, and a spilled mask is a hazard (gl.hpp).
__device__ __forceinline__ void sbox_all(uint64_t (&s)[12]) {
#pragma unroll
  for (int g = 0; g < 4; g++) {
    uint64_t y[3] = {s[3 * g], s[3 * g + 1], s[3 * g + 2]};
    sbox_n<3>(y);
#pragma unroll
    for (int i = 0; i < 3; i++) s[3 * g + i] = y[i];
    __builtin_amdgcn_sched_barrier(0);  // one group's masks at a time: the scheduler otherwise overlaps two groups and spills
  }
}

using mxa::reduce_rows;  // mx_arith.cuh: (L, H) accumulator pairs -> reduced words

// MDS = circulant(17,15,41,16,2,28,13,13,39,18,34,20) + diag(8,0,...).  Entries are < 2^6, so the
// 32-bit halves of the state are accumulated separately in 64 bits (no overflow: 12*41*2^32 plus a
// 32-bit constant) and recombined with one small reduction per row.  `rc_next` (nullable) points at
// the NEXT round's 12 constants: they are added inside the accumulators (the multiply-add's addend),
// so a round never pays a separate modular addition for them.
template <bool ADD_RC>
__device__ __forceinline__ void mds(uint64_t (&s)[12], const uint64_t* __restrict__ rc_next) {
  constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  uint32_t lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    lo[i] = (uint32_t)s[i];
    hi[i] = (uint32_t)(s[i] >> 32);
  }
#pragma unroll
  for (int g = 0; g < 3; g++) {
    uint64_t L[4], H[4];
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const int r = 4 * g + rr;
      L[rr] = 0;
      H[rr] = 0;
      if (ADD_RC) {
        const uint64_t k = rc_next[r];
        L[rr] = (uint32_t)k;
        H[rr] = k >> 32;
      }
#pragma unroll
      for (int i = 0; i < 12; i++) {
        L[rr] += (uint64_t)lo[(i + r) % 12] * C[i];
        H[rr] += (uint64_t)hi[(i + r) % 12] * C[i];
      }
      if (r == 0) {
        L[rr] += (uint64_t)lo[0] * 8;
        H[rr] += (uint64_t)hi[0] * 8;
      }
    }
    uint64_t out[4];
    reduce_rows<4>(L, H, out);
#pragma unroll
    for (int rr = 0; rr < 4; rr++) s[4 * g + rr] = out[rr];
  }
}

__device__ __forceinline__ void permute(uint64_t (&s)[12]) {
  // round r: (state + RC[r]) -> S-box -> MDS; RC[r+1] rides in round r's MDS accumulators
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl::add(s[i], RC[i]);
  int rnd = 0;
#pragma unroll 1
  for (int k = 0; k < 4; k++, rnd++) {
    sbox_all(s);
    mds<true>(s, RC + (rnd + 1) * 12);
  }
#pragma unroll 1
  for (int k = 0; k < 22; k++, rnd++) {
    s[0] = sbox(s[0]);
    mds<true>(s, RC + (rnd + 1) * 12);
  }
#pragma unroll 1
  for (int k = 0; k < 3; k++, rnd++) {
    sbox_all(s);
    mds<true>(s, RC + (rnd + 1) * 12);
  }
  sbox_all(s);
  mds<false>(s, nullptr);
}

// ------------------------------------------------------------------------------------------------
// Quad-cooperative permutation: FOUR lanes own one state, lane q (= lane & 3) holds words q, q+4,
// q+8.  The MDS inputs of the other three lanes arrive through DPP quad_perm broadcasts (register
// to register, no LDS), each lane then produces its three output rows.  Compared with one lane per
// state this has 4x the waves for the same number of permutations and ~2.5x lower latency per
// permutation -- what medium-sized launches (2^13..2^17 rows, Merkle upper levels) need to fill
// 1024 SIMDs -- at ~1.6x the total instruction count (three of four lanes idle in the partial-round
// S-box).  Large launches keep the one-lane form.
// ------------------------------------------------------------------------------------------------
template <int P>
__device__ __forceinline__ uint64_t quad_bcast(uint64_t v) {
  constexpr int ctrl = P | (P << 2) | (P << 4) | (P << 6);  // quad_perm:[P,P,P,P]
  const int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)v, ctrl, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(v >> 32), ctrl, 0xF, 0xF, true);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

struct QuadCtx {
  uint32_t cf[12];  // cf[k] = C[(k - q) mod 12]: coefficient of input k in output row q (rows q+4a rotate by 4a)
  uint32_t diag;    // 8 on lane q == 0 (row 0), else 0
  uint32_t q;
};
__device__ __forceinline__ QuadCtx quad_ctx() {
  constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  QuadCtx c;
  c.q = threadIdx.x & 3;
#pragma unroll
  for (int k = 0; k < 12; k++) {
    uint32_t v = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) v = (c.q == (uint32_t)q) ? C[(k - q + 12) % 12] : v;
    c.cf[k] = v;
  }
  c.diag = c.q == 0 ? 8u : 0u;
  return c;
}

// e[a] = state word q + 4a.  rc: the 360 round constants in LDS or global memory.
__device__ __forceinline__ void permute_quad(uint64_t (&e)[3], const QuadCtx& c, const uint64_t* __restrict__ rc) {
  {
    const uint64_t* r0 = rc + c.q;
#pragma unroll
    for (int a = 0; a < 3; a++) e[a] = gl::add(e[a], r0[4 * a]);
  }
#pragma unroll 1
  for (int rnd = 0; rnd < 30; rnd++) {
    const bool full = rnd < 4 || rnd >= 26;
    // next round's constants for this lane's three rows (zero after the last round); they are added
    // inside the MDS accumulators
    const uint64_t* r = rc + (rnd < 29 ? rnd + 1 : 0) * 12 + c.q;
    const uint64_t kmask = rnd < 29 ? ~0ULL : 0ULL;
    if (full) {
      sbox_n<3>(e);
    } else {
      const uint64_t sb = sbox(e[0]);  // only state word 0 (lane q == 0, slot 0) takes it
      e[0] = c.q == 0 ? sb : e[0];
    }
    // The DPP broadcasts below read registers that asm statements have just written (the S-boxes above,
    // and in partial rounds e[1], e[2] straight from the previous round's reduce_rows): a DPP read needs
    // 2 wait states after a VALU write of its source, and the hazard recogniser does not see writes made
    // inside inline asm.  The dependence on e[] pins this nop between the two on every path.
    asm volatile("s_nop 1" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]));
    // gather the whole state: b[p + 4a] = word (p + 4a), held by lane p
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const uint64_t b0 = quad_bcast<0>(e[a]), b1 = quad_bcast<1>(e[a]), b2 = quad_bcast<2>(e[a]),
                     b3 = quad_bcast<3>(e[a]);
      lo[4 * a] = (uint32_t)b0; hi[4 * a] = (uint32_t)(b0 >> 32);
      lo[4 * a + 1] = (uint32_t)b1; hi[4 * a + 1] = (uint32_t)(b1 >> 32);
      lo[4 * a + 2] = (uint32_t)b2; hi[4 * a + 2] = (uint32_t)(b2 >> 32);
      lo[4 * a + 3] = (uint32_t)b3; hi[4 * a + 3] = (uint32_t)(b3 >> 32);
    }
    uint64_t L[3], H[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {  // output row q + 4a
      const uint64_t k = r[4 * a] & kmask;
      L[a] = (uint32_t)k;
      H[a] = k >> 32;
#pragma unroll
      for (int kk = 0; kk < 12; kk++) {
        const uint32_t cf = c.cf[(kk - 4 * a + 12) % 12];
        L[a] += (uint64_t)lo[kk] * cf;
        H[a] += (uint64_t)hi[kk] * cf;
      }
      if (a == 0) {
        L[a] += (uint64_t)lo[0] * c.diag;
        H[a] += (uint64_t)hi[0] * c.diag;
      }
    }
    reduce_rows<3>(L, H, e);
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
