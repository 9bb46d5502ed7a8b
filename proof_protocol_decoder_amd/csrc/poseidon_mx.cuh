// poseidon_mx.cuh -- Poseidon-Goldilocks with the MDS layer on the matrix cores (third kernel form, "mx").
//
// Same permutation as poseidon.cuh (upstream plonky2::hash::poseidon, reached from
// plonky_block_proof_gen/src/proof_gen.rs:44-52 through MerkleTree::new), bit for bit.  Why a matrix-core form of an
// integer hash: the field kernels are bound by VALU issue, and in the one-lane form the MDS layer is more than half
// of the instructions of a permutation (288 32x6-bit multiply-adds + 48 for the row reductions per round, against 60
// per S-box).  The MDS layer is a genuine matrix product -- a constant 12x12 matrix of 6-bit entries times the state
// -- and it is exact on the int8 MFMA: split every state word into its 8 BYTES (the register bytes as they are: no
// data movement), multiply each byte plane by the matrix (products < 2^14, row sums < 2^17 in the i32 accumulators),
// and recombine the 8 plane sums of a word with shifts: y = sum_p a_p * 2^(8p) (mod p).  That replaces 28 VALU
// instructions per output word by 12, and the MFMA pipe was idle.  An MFMA is not free next to VALU work on this
// chip (~10 SIMD cycles each, tools/mfma_probe.hip), but one replaces about twenty multiply-adds here.
//
// Layout.  A wave owns 16 * NS states as NS (4, 2 or 1) "sets" of 16.  Lane l = (n = l & 15, kb = l >> 4) holds, for
// every set m, words kb, kb + 4, kb + 8 of state 16m + n: e[m][a] = word kb + 4a.  This is exactly the operand map of
// v_mfma_i32_16x16x64_i8 (checked on the device, tools/mfma_probe.hip): B[k][col]: lane (col = l & 15, k-block l >> 4)
// supplies 16 bytes, A[row][k] likewise with row = l & 15, products are paired by (k-block, byte), and the result
// D[row][col] lands in lane (col, row >> 2), register row & 3.
//   B operand of (set m, half h): dwords { half h of e[m][0], of e[m][1], of e[m][2], 0 } ^ 0x80808080: byte 4a + pp is
//     plane 4h + pp of input word kb + 4a.  The XOR makes the byte x the signed x - 128 the MFMA multiplies.
//   A operand of output slot g (constant, 3 x 4 VGPRs): lane (r, kb), dword a = M[(r >> 2) + 4g][kb + 4a] << 8(r & 3):
//     row r of the tile is (output word (r >> 2) + 4g, plane r & 3 of the half) and takes only that plane's bytes.
//   C operand: 128 * rowsum (undoes the -128) + the matching byte of the NEXT round's constant: the constant layer
//     rides in the accumulators exactly as in the one-lane form.  30 x 4 x 24 dwords, a compile-time table each
//     workgroup copies into LDS.
//   D of (g, h): lane (n, ib) register reg = plane 4h + reg of output word ib + 4g of state n -- the layout the state
//     had: no lane ever moves data in the MDS layer.
// Six MFMAs per set and round.  Partial rounds: word 0 of the four sets sits in lanes 0..15 of four registers; three
// v_permlane{16,32}_swap per 32-bit half gather them into one dense register (all 64 lanes busy in the S-box) and the
// same three swaps, being involutions, put everything back.  NS = 2 and 1 (32 and 16 states per wave: 2x and 4x the
// waves for a launch too small to fill 1024 SIMDs otherwise) use one swap or none and leave lanes idle in that S-box.
#pragma once
#include "poseidon.cuh"
#include "poseidon_group.hpp"

namespace poseidon {
namespace mx {

using mxa::v4i;

constexpr int CIN_PER_ROUND = 4 * 24;            // [ib][2g + h][reg]
constexpr int CIN_WORDS = 30 * CIN_PER_ROUND;    // 11,520 bytes of LDS

__device__ __forceinline__ uint32_t mds_entry(uint32_t i, uint32_t k) {  // M[i][k] = C[(k - i) mod 12] (+ 8 at [0][0])
  constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  const uint32_t d = (k + 12 - i) % 12;
  uint32_t v = 0;
#pragma unroll
  for (int j = 0; j < 12; j++) v = d == (uint32_t)j ? C[j] : v;
  return v + ((i | k) == 0 ? 8u : 0u);
}

// The C-operand table, made at compile time: [round][ib][2g + h][reg] = 128 * rowsum(word ib + 4g) + byte 4h + reg of
// the NEXT round's constant of that word (none after the last round).
struct CinTable {
  uint32_t v[CIN_WORDS];
};
constexpr CinTable make_cin_table() {
  constexpr uint64_t rc[360] = {
#include "poseidon_rc.inc"
  };
  CinTable t{};
  for (int i = 0; i < CIN_WORDS; i++) {
    const int rnd = i / CIN_PER_ROUND, rem = i % CIN_PER_ROUND, ib = rem / 24, idx = rem % 24;
    const int g = idx >> 3, h = (idx >> 2) & 1, reg = idx & 3, wo = ib + 4 * g;
    uint32_t v = 128u * (256u + (wo == 0 ? 8u : 0u));
    if (rnd < 29) v += (uint32_t)(rc[(rnd + 1) * 12 + wo] >> (8 * (4 * h + reg))) & 0xFFu;
    t.v[i] = v;
  }
  return t;
}
static __device__ const CinTable CIN_TABLE __attribute__((aligned(16))) = make_cin_table();

// the whole workgroup copies the table into LDS, 16 bytes per lane and step (call once, then __syncthreads)
__device__ __forceinline__ void build_cin(uint32_t* __restrict__ cin) {
  const uint4* src = (const uint4*)CIN_TABLE.v;
  uint4* dst = (uint4*)cin;
  for (uint32_t i = threadIdx.x; i < (uint32_t)CIN_WORDS / 4; i += blockDim.x) dst[i] = src[i];
}

struct Ctx {
  v4i A[3];          // the MDS matrix as the A operand of output slot g
  uint32_t kb;       // lane >> 4: this lane holds words kb, kb + 4, kb + 8
  const uint32_t* cin;  // LDS table, already offset to this lane's row group
};
__device__ __forceinline__ Ctx make_ctx(const uint32_t* cin_lds) {
  Ctx c;
  const uint32_t lane = threadIdx.x & 63, r = lane & 15;
  c.kb = lane >> 4;
#pragma unroll
  for (int g = 0; g < 3; g++) {
#pragma unroll
    for (int a = 0; a < 3; a++) c.A[g][a] = (int)(mds_entry((r >> 2) + 4 * g, c.kb + 4 * a) << (8 * (r & 3)));
    c.A[g][3] = 0;
  }
  c.cin = cin_lds + c.kb * 24;
  return c;
}

using mxa::planes;  // four plane sums -> a0 + a1*2^8 + a2*2^16 + a3*2^24

// MDS layer (+ next round's constants) of every set
template <int NS>
__device__ __forceinline__ void mds(uint64_t (&e)[NS][3], const Ctx& c, int rnd) {
  const v4i* cr = (const v4i*)(c.cin + rnd * CIN_PER_ROUND);
  v4i cin[6];
#pragma unroll
  for (int t = 0; t < 6; t++) cin[t] = cr[t];
#pragma unroll
  for (int m = 0; m < NS; m++) {
    v4i blo, bhi;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      blo[a] = (int)((uint32_t)e[m][a] ^ 0x80808080u);
      bhi[a] = (int)((uint32_t)(e[m][a] >> 32) ^ 0x80808080u);
    }
    blo[3] = 0;
    bhi[3] = 0;
    uint64_t L[3], H[3];
#pragma unroll
    for (int g = 0; g < 3; g++) {
      const v4i dl = __builtin_amdgcn_mfma_i32_16x16x64_i8(c.A[g], blo, cin[2 * g], 0, 0, 0);
      const v4i dh = __builtin_amdgcn_mfma_i32_16x16x64_i8(c.A[g], bhi, cin[2 * g + 1], 0, 0, 0);
      L[g] = planes(dl);
      H[g] = planes(dh);
    }
    reduce_rows<3>(L, H, e[m]);
  }
}

// S-box on state word 0 only (partial rounds)
template <int NS>
__device__ __forceinline__ void sbox_word0(uint64_t (&e)[NS][3], const Ctx& c) {
  if constexpr (NS == 4) {
    uint32_t l0 = (uint32_t)e[0][0], l1 = (uint32_t)e[1][0], l2 = (uint32_t)e[2][0], l3 = (uint32_t)e[3][0];
    uint32_t h0 = (uint32_t)(e[0][0] >> 32), h1 = (uint32_t)(e[1][0] >> 32), h2 = (uint32_t)(e[2][0] >> 32),
             h3 = (uint32_t)(e[3][0] >> 32);
    // rows (16 lanes) of x0 become [x0.r0, x1.r0, x2.r0, x3.r0]: word 0 of all 64 states.  The operands come from asm
    // statements (reduce_rows), which the hazard recogniser does not see: 2 wait states before every swap that reads a
    // register an earlier instruction has just written.
    asm("s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
        "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
        "s_nop 0\n\t"
        "v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %4, %6\n\ts_nop 1"
        : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3));
    const uint64_t y = sbox(gl::cc::mk64(l0, h0));
    l0 = (uint32_t)y;
    h0 = (uint32_t)(y >> 32);
    asm("s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %4, %6\n\t"
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
        "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\ts_nop 1"
        : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3));
    e[0][0] = gl::cc::mk64(l0, h0);
    e[1][0] = gl::cc::mk64(l1, h1);
    e[2][0] = gl::cc::mk64(l2, h2);
    e[3][0] = gl::cc::mk64(l3, h3);
  } else if constexpr (NS == 2) {
    // one swap per half puts word 0 of both sets into lanes 0..31; lanes 32..63 (words 2 of the two sets) keep theirs
    uint32_t l0 = (uint32_t)e[0][0], l1 = (uint32_t)e[1][0], h0 = (uint32_t)(e[0][0] >> 32), h1 = (uint32_t)(e[1][0] >> 32);
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
        : "+v"(l0), "+v"(l1), "+v"(h0), "+v"(h1));
    const uint64_t y = sbox(gl::cc::mk64(l0, h0));
    const bool word0 = c.kb < 2;
    l0 = word0 ? (uint32_t)y : l0;
    h0 = word0 ? (uint32_t)(y >> 32) : h0;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
        : "+v"(l0), "+v"(l1), "+v"(h0), "+v"(h1));
    e[0][0] = gl::cc::mk64(l0, h0);
    e[1][0] = gl::cc::mk64(l1, h1);
  } else {
    static_assert(NS == 1, "1, 2 or 4 sets per wave");
    const uint64_t y = sbox(e[0][0]);
    e[0][0] = c.kb == 0 ? y : e[0][0];
  }
}

template <int NS>
__device__ __forceinline__ void sbox_full(uint64_t (&e)[NS][3]) {
#pragma unroll
  for (int m = 0; m < NS; m++) {
    sbox_n<3>(e[m]);
    __builtin_amdgcn_sched_barrier(0);  // one group's carry masks at a time (poseidon.cuh, sbox_all)
  }
}

// ---- grouped partial rounds (four sets per wave only) -----------------------------------------------------------
// Up to K = 8 partial rounds at a time (poseidon_group.hpp; integer model tools/poseidon_group_model.py): within a
// group only ONE word per state and round -- the next S-box input, an affine form of the eleven untouched words and the
// earlier S-box outputs -- comes out of the matrix cores and is recombined; the twelve words are recombined once per
// group.  228 -> ~140 VALU instructions per round and 64 states, 6 -> 5 MFMAs per set and round.  The 22 partial
// rounds are 8 + 8 + 6: the third group runs steps 0..5 on the same form operands.
//   operands (1 KiB each, shared by the four sets): W = forms over the bytes of w, D = one more sigma into the
//   forms still to come, MAIN = the new state over (w, sigma_0..7) -- 40 operands in LDS, identical for the two long
//   groups --, and 18 MAIN operands of the short group in global memory; C tables per group hold the round constants.
//   B operands: blo / bhi = the state's byte planes as in mds(); bsig: lane group g holds sigma_g and sigma_{g+4}.
//   Form f of a pair comes out in lane group f % 4 (rows 4(f%4)..+3 of the tile), so the gather into the dense
//   S-box register and the way back come in four variants of the same three permlane swaps.
namespace grp {
constexpr int K = 8;
constexpr poseidon::group::Layout LAY = poseidon::group::layout(K);
// NG = number of groups: 2 = partial rounds 4..11 and 12..19, rounds 20..25 in the per-round form;
//      3 = also rounds 20..25, as a SHORT group: steps 0..5 on the same form operands (the rows of forms 6 and 7 are
//          garbage nobody reads), the new state from the MAIN operands of a six-round group (18 more operands,
//          read from global memory).
constexpr int SHORT_K = 6;
constexpr int OPS_WORDS = LAY.n_ops * 256;        // 40 KiB
constexpr int C_WORDS = poseidon::group::CFORM_WORDS + poseidon::group::CMAIN_WORDS;
constexpr int MDS_A_WORDS = 3 * 256;              // the per-round MDS layer's three A operands (Ctx::A), read from LDS
                                                  // per round instead of living in 12 VGPRs through the groups
// device image: group operands, per group cform + cmain, the MDS layer's A operands -- this much goes to LDS --, then
// (NG = 3) the short group's 18 MAIN operands, which stay in global memory (L2): with them in LDS a workgroup needs
// 67 KB, two per CU, and the kernels lose a quarter of their resident waves (profiles/r3_poseidon_three_groups.txt)
template <int NG> constexpr int TABLE_WORDS = OPS_WORDS + NG * C_WORDS + MDS_A_WORDS;
template <int NG> constexpr int IMAGE_WORDS = TABLE_WORDS<NG> + (NG == 3 ? 18 * 256 : 0);

// the whole workgroup copies the image into LDS, 16 bytes per lane and step (call once, then __syncthreads)
template <int NG>
__device__ __forceinline__ void load_tables(uint32_t* __restrict__ lds, const uint32_t* __restrict__ glob) {
  const uint4* src = (const uint4*)glob;
  uint4* dst = (uint4*)lds;
  for (uint32_t i = threadIdx.x; i < (uint32_t)TABLE_WORDS<NG> / 4; i += blockDim.x) dst[i] = src[i];
}

// gather<F>: x[m] holds set m's value in lane group F; afterwards x[F] holds set j's value in lane group j.
// The same instructions in reverse order undo it (each swap is an involution).  Operands come from / go to asm
// statements the hazard recogniser does not see: 2 wait states around every swap by hand.
template <int F>
__device__ __forceinline__ void gather(uint32_t (&l)[4], uint32_t (&h)[4]) {
  if constexpr ((F & 1) == 0)
    asm("s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
        "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
        "s_nop 0\n\t"
        "v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %4, %6\n\ts_nop 1"
        : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]));
  else
    asm("s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
        "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
        "s_nop 0\n\t"
        "v_permlane32_swap_b32 %1, %3\n\tv_permlane32_swap_b32 %5, %7\n\ts_nop 1"
        : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]));
}
template <int F>
__device__ __forceinline__ void scatter(uint32_t (&l)[4], uint32_t (&h)[4]) {
  if constexpr ((F & 1) == 0)
    asm("s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %4, %6\n\t"
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
        "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\ts_nop 1"
        : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]));
  else
    asm("s_nop 1\n\t"
        "v_permlane32_swap_b32 %1, %3\n\tv_permlane32_swap_b32 %5, %7\n\t"
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
        "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\ts_nop 1"
        : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]));
}

struct State {
  v4i blo[4], bhi[4], bsig[4];  // B operands of the four sets (bytes ^ 0x80)
  v4i acc[4][2];                // the live form pair: tile L / H of every set
};

// step J of a group: S-box input J -> sigma_J into bsig and into the forms still to come
// (full = false: the short group, whose last step is 5: no later form takes sigma_5)
template <int J>
__device__ __forceinline__ void step(State& s, uint64_t (&e)[4][3], const v4i* ops, const int* cform, uint32_t lane,
                                     bool full = true) {
  constexpr int P = J / 4, F = J % 4;
  const uint32_t kb = lane >> 4;
  if constexpr (F == 0) {  // start pair P: constants + the forms over w (+ the sigmas known so far)
    constexpr int nw = LAY.w_per_half[P];
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const v4i c0 = ((const v4i*)(cform + (P * 2 + half) * 16))[kb];
      const v4i a_lo = ops[(LAY.w_base[P] + half * nw) * 64 + lane], a_hi = ops[(LAY.w_base[P] + half * nw + 1) * 64 + lane];
#pragma unroll
      for (int m = 0; m < 4; m++) {
        s.acc[m][half] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_lo, s.blo[m], c0, 0, 0, 0);
        s.acc[m][half] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_hi, s.bhi[m], s.acc[m][half], 0, 0, 0);
      }
      if constexpr (P > 0) {
        const v4i a_sg = ops[(LAY.w_base[P] + half * nw + 2) * 64 + lane];
#pragma unroll
        for (int m = 0; m < 4; m++) s.acc[m][half] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a_sg, s.bsig[m], s.acc[m][half], 0, 0, 0);
      }
    }
  }
  uint32_t l[4], h[4];
  if constexpr (J == 0) {  // F_0 is the state's own word 0 (lane group 0)
#pragma unroll
    for (int m = 0; m < 4; m++) { l[m] = (uint32_t)e[m][0]; h[m] = (uint32_t)(e[m][0] >> 32); }
  } else {                 // form J: planes 0..3 / 4..7 in this lane's registers of tile L / H (lane group F)
    uint64_t L[4], H[4], x[4];
#pragma unroll
    for (int m = 0; m < 4; m++) { L[m] = planes(s.acc[m][0]); H[m] = planes(s.acc[m][1]); }
    reduce_rows<4>(L, H, x);
#pragma unroll
    for (int m = 0; m < 4; m++) { l[m] = (uint32_t)x[m]; h[m] = (uint32_t)(x[m] >> 32); }
  }
  gather<F>(l, h);
  const uint64_t y = sbox(gl::cc::mk64(l[F], h[F])) ^ 0x8080808080808080ULL;  // sigma_J of all 64 states, as a B operand
  l[F] = (uint32_t)y;
  h[F] = (uint32_t)(y >> 32);
  scatter<F>(l, h);        // set m's sigma_J in lane group F of (l[m], h[m])
#pragma unroll
  for (int m = 0; m < 4; m++) {  // ... into its slot of bsig: only row F of the wave is written
    s.bsig[m][2 * (J / 4)] = __builtin_amdgcn_update_dpp(s.bsig[m][2 * (J / 4)], (int)l[m], 0xE4, 1 << F, 0xF, false);
    s.bsig[m][2 * (J / 4) + 1] = __builtin_amdgcn_update_dpp(s.bsig[m][2 * (J / 4) + 1], (int)h[m], 0xE4, 1 << F, 0xF, false);
  }
  if constexpr (J >= LAY.d_first[P] && J < LAY.d_first[P] + LAY.d_count[P]) {  // a later form of the pair needs sigma_J
    if (J == SHORT_K - 1 && !full) return;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const v4i a = ops[(LAY.d_base[P] + 2 * (J - LAY.d_first[P]) + half) * 64 + lane];
#pragma unroll
      for (int m = 0; m < 4; m++) s.acc[m][half] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, s.bsig[m], s.acc[m][half], 0, 0, 0);
    }
  }
}

// rounds r0 .. r0 + 7 of the partial rounds: e = t(r0) in, t(r0 + 8) out (S-box-input form, constants included);
// the third group of NG = 3 is short: rounds 20..25, t(26) out
template <int NG>
__device__ __forceinline__ void partial_group(uint64_t (&e)[4][3], const uint32_t* tab, const uint32_t* __restrict__ gtab, int grp) {
  const uint32_t lane = threadIdx.x & 63, kb = lane >> 4;
  const bool full = NG == 2 || grp < 2;   // wave-uniform
  const v4i* ops = (const v4i*)tab;
  const int* cform = (const int*)(tab + OPS_WORDS + grp * C_WORDS);
  const int* cmain = cform + poseidon::group::CFORM_WORDS;
  State s;
#pragma unroll
  for (int m = 0; m < 4; m++) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      s.blo[m][a] = (int)((uint32_t)e[m][a] ^ 0x80808080u);
      s.bhi[m][a] = (int)((uint32_t)(e[m][a] >> 32) ^ 0x80808080u);
    }
    s.blo[m][3] = 0;
    s.bhi[m][3] = 0;
    s.bsig[m] = v4i{0, 0, 0, 0};
  }
  step<0>(s, e, ops, cform, lane);
  step<1>(s, e, ops, cform, lane);
  step<2>(s, e, ops, cform, lane);
  step<3>(s, e, ops, cform, lane);
  step<4>(s, e, ops, cform, lane);
  step<5>(s, e, ops, cform, lane, full);
  if (full) {
    step<6>(s, e, ops, cform, lane);
    step<7>(s, e, ops, cform, lane);
  }
  // the new state: twelve words over (w, sigma_0 .. sigma_7 / sigma_5), recombined as in mds()
  const v4i* lops = ops + LAY.main_base * 64 + lane;
  const v4i* gops = (const v4i*)(gtab + TABLE_WORDS<NG>) + lane;
#pragma unroll
  for (int g = 0; g < 3; g++) {
    v4i d[4][2], A[2][3];
    if (full) {
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int c = 0; c < 3; c++) A[h][c] = lops[((g * 2 + h) * 3 + c) * 64];
    } else {
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int c = 0; c < 3; c++) A[h][c] = gops[((g * 2 + h) * 3 + c) * 64];
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const v4i c0 = ((const v4i*)(cmain + (g * 2 + h) * 16))[kb];
#pragma unroll
      for (int m = 0; m < 4; m++) {
        d[m][h] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[h][0], s.blo[m], c0, 0, 0, 0);
        d[m][h] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[h][1], s.bhi[m], d[m][h], 0, 0, 0);
        d[m][h] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[h][2], s.bsig[m], d[m][h], 0, 0, 0);
      }
    }
    uint64_t L[4], H[4], x[4];
#pragma unroll
    for (int m = 0; m < 4; m++) { L[m] = planes(d[m][0]); H[m] = planes(d[m][1]); }
    reduce_rows<4>(L, H, x);
#pragma unroll
    for (int m = 0; m < 4; m++) e[m][g] = x[m];
  }
}
}  // namespace grp

// The grouped kernels keep the C table of the per-round MDS layer only for the rounds that still use it: 0..3 and
// 20..29 (NG = 2: 14 x 384 bytes instead of 30 x 384: with the 45 KiB of group operands three workgroups still fit a
// CU's LDS) or 0..3 and 26..29 (NG = 3)
template <int NG> constexpr int CIN_GROUPED_ROUNDS = NG == 3 ? 8 : 14;
template <int NG> constexpr int CIN_GROUPED_WORDS = CIN_GROUPED_ROUNDS<NG> * CIN_PER_ROUND;
template <int NG> __device__ __forceinline__ int cin_slot(int rnd) { return rnd < 4 ? rnd : rnd - (30 - CIN_GROUPED_ROUNDS<NG>); }
template <int NG>
__device__ __forceinline__ void build_cin_grouped(uint32_t* __restrict__ cin) {
  const uint4* src = (const uint4*)CIN_TABLE.v;
  uint4* dst = (uint4*)cin;
  for (uint32_t i = threadIdx.x; i < (uint32_t)CIN_GROUPED_WORDS<NG> / 4; i += blockDim.x)
    dst[i] = src[i < 4 * CIN_PER_ROUND / 4 ? i : i + (30 - CIN_GROUPED_ROUNDS<NG>) * CIN_PER_ROUND / 4];
}

// mds<4> with the A operands read from the LDS image (one ds_read_b128 each per round, shared by the four sets)
template <int NG>
__device__ __forceinline__ void mds4_lds(uint64_t (&e)[4][3], const Ctx& c, int rnd, const uint32_t* tab) {
  const v4i* cr = (const v4i*)(c.cin + cin_slot<NG>(rnd) * CIN_PER_ROUND);
  const v4i* am = (const v4i*)(tab + grp::OPS_WORDS + NG * grp::C_WORDS) + (threadIdx.x & 63);
#pragma unroll
  for (int m = 0; m < 4; m++) {
    v4i blo, bhi;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      blo[a] = (int)((uint32_t)e[m][a] ^ 0x80808080u);
      bhi[a] = (int)((uint32_t)(e[m][a] >> 32) ^ 0x80808080u);
    }
    blo[3] = 0;
    bhi[3] = 0;
    uint64_t L[3], H[3];
#pragma unroll
    for (int g = 0; g < 3; g++) {
      const v4i A = am[g * 64];
      const v4i dl = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, blo, cr[2 * g], 0, 0, 0);
      const v4i dh = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, bhi, cr[2 * g + 1], 0, 0, 0);
      L[g] = planes(dl);
      H[g] = planes(dh);
    }
    reduce_rows<3>(L, H, e[m]);
  }
}

// The permutation with the partial rounds grouped (four sets per wave; tab = the LDS image of grp::load_tables<NG>):
// NG = 2: rounds 4..19 in two groups, 20..25 one by one; NG = 3: all 22 partial rounds in groups (8 + 8 + 6)
template <int NG>
__device__ __forceinline__ void permute_grouped(uint64_t (&e)[4][3], const Ctx& c, const uint32_t* tab,
                                                const uint32_t* __restrict__ gtab) {
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const uint64_t k = RC[c.kb + 4 * a];
#pragma unroll
    for (int m = 0; m < 4; m++) e[m][a] = gl::add(e[m][a], k);
  }
  int rnd = 0;
#pragma unroll 1
  for (int k = 0; k < 4; k++, rnd++) {
    sbox_full<4>(e);
    mds4_lds<NG>(e, c, rnd, tab);
  }
#pragma unroll 1
  for (int g = 0; g < NG; g++) grp::partial_group<NG>(e, tab, gtab, g);
  rnd = NG == 3 ? 26 : 20;
  if constexpr (NG == 2) {
#pragma unroll 1
    for (; rnd < 26; rnd++) {
      sbox_word0<4>(e, c);
      mds4_lds<NG>(e, c, rnd, tab);
    }
  }
#pragma unroll 1
  for (int k = 0; k < 4; k++, rnd++) {
    sbox_full<4>(e);
    mds4_lds<NG>(e, c, rnd, tab);
  }
}

// e[m][a] = word kb + 4a of state 16m + n, any u64 in, reduced out
template <int NS>
__device__ __forceinline__ void permute(uint64_t (&e)[NS][3], const Ctx& c) {
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const uint64_t k = RC[c.kb + 4 * a];
#pragma unroll
    for (int m = 0; m < NS; m++) e[m][a] = gl::add(e[m][a], k);
  }
  int rnd = 0;
#pragma unroll 1
  for (int k = 0; k < 4; k++, rnd++) {
    sbox_full<NS>(e);
    mds<NS>(e, c, rnd);
  }
#pragma unroll 1
  for (int k = 0; k < 22; k++, rnd++) {
    sbox_word0<NS>(e, c);
    mds<NS>(e, c, rnd);
  }
#pragma unroll 1
  for (int k = 0; k < 4; k++, rnd++) {
    sbox_full<NS>(e);
    mds<NS>(e, c, rnd);
  }
}

}  // namespace mx
}  // namespace poseidon
