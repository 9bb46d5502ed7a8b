// ntt_mx.cuh -- K2 with the butterflies on the matrix cores: the LDS-block kernels (2^12..2^14 points) as three
// radix-16 passes whose 16-point DFTs are int8 MFMAs.  Included by ntt.hip (which owns Ntt16Args and the planner).
//
// Restates plonky2_field::fft / ifft / coset_fft like ntt.hip (reached from plonky_block_proof_gen/src/
// proof_gen.rs:44-52 through PolynomialBatch::from_values); results are bit-identical to the VALU kernels.
//
// Why: the VALU kernels are bound by instruction issue (a radix-2 butterfly is 27 instructions, 15 of them the
// modular multiply: ~214 per element for 2^14 points).  In Goldilocks 2 is a 192nd root of unity and the 16th root
// of the plonky2 generator is w16 = 2^156 = -2^60, so EVERY entry w16^(jk) * 2^(8p) of "16-point DFT acting on the
// bytes of its inputs" is +-2^e or +-(2^a - 2^b): written in balanced base-256 digits it is an int8 matrix
//     A[(k, q)][(j, p)] = digit q of (w16^(jk) * 2^(8p) mod p),      y_k = sum_q 2^(8q) * sum_(j,p) A * byte p of x_j,
// 128 x 128, exact in the i32 accumulators (|sum| < 2^21).  One pass = 16 v_mfma_i32_16x16x64_i8 per 256 elements
// (one per 16 elements; each blocks its SIMD for ~12 cycles, tools/mfma_probe.hip) + per element: 2 XORs (byte x -> signed x - 128), the
// shift recombination of 8 plane sums (12 instructions, mx_arith.cuh) and ONE modular multiply by the inter-pass
// twiddle: ~30 instructions per element and pass against 4 x 13.5 for four radix-2 stages.  tools/ntt_mx_model.py is
// the integer model of all of this (digits, bias, pass structure, LDS swizzle).
//
// Layout of a pass.  A tile = 16 groups of 16 elements; lane (n = lane & 15, kb = lane >> 4) supplies four elements of
// group n -- its registers as they are, XOR 0x80808080 -- two in K-chunk 0 and the two with index + 8 in chunk 1, and
// receives four outputs, digits 4h + reg of row block 2a + h (a = 0..3).  Row constants (C operand): a bias that is a
// multiple of p and makes every plane sum non-negative, plus the correction for the -128.  A wave does four tiles
// ("sets") per pass: 16 elements per lane, like the VALU kernels.
//   decimation in frequency (natural -> bit-reversed): pass S = 4096, 256, 16 on position blk*S + j*(S/16) + i:
//     y_k = DFT16(x_j), times w_S^(i k), stored at blk*S + bitrev4(k)*(S/16) + i; slot (chunk c, element eps) of lane
//     kb is input j = 8c + 2kb + eps, row a of lane ib is output k = 4 ib + a;
//   decimation in time (bit-reversed -> natural): the transposed pipeline, S = 16, 256, 4096, twiddle before the DFT;
//     slot = input kin = 8c + 4 eps + kb (found at field bitrev4(kin)), row = output j = (a & 1) + 2 ib + 8 (a >> 1).
//   With these assignments every LDS access of a pass is bank-conflict free under one XOR swizzle (swz12; checked
//   exhaustively by tools/ntt_mx_model.py), chunk 1 is "chunk 0's input + 8", and all rows of a tile have outputs of
//   one parity (a & 1) -- which is what lets chunk 1 reuse chunk 0's A registers (Frag, below).
// 2^13 / 2^14-point blocks: one / two radix-2 stages (VALU, the twiddles of ntt.hip) join 2 / 4 sub-blocks of 4096,
// done in registers next to the outermost pass, which reads from / writes to global memory directly (128-byte runs).
#pragma once
#include "mx_arith.cuh"

namespace mxn {

using mxa::v4i;

// device image of the constants of one (kind, direction): kind 0 = DIF matrix, 1 = DIT matrix
struct Tables {
  uint32_t a[8 * 64 * 4];       // A operand of K-chunk 0: [row block][lane] x 16 bytes (chunk 1 = +-the same, see dft16)
  uint32_t c[8 * 4 * 4];        // C operand: [row block][ib] x 4 i32
  uint64_t tw256[16 * 256];     // w_4096^(i k), [k][i]
  uint64_t tw16[16 * 16];       // w_256^(i k), [k][i]
};
constexpr int C_LDS_WORDS = 8 * 4 * 4;   // the C operands live in LDS behind the data image (512 bytes)

__device__ __forceinline__ uint32_t swz12(uint32_t pos) {  // XOR swizzle of a 4096-element LDS image (8-byte words)
  return pos ^ ((pos >> 4) & 15u) ^ ((((pos >> 1) ^ (pos >> 3) ^ (pos >> 9) ^ (pos >> 11)) & 1u) << 4);
}
__host__ __device__ __forceinline__ constexpr uint32_t br4(uint32_t k) {
  return ((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3);
}

// Register budget.  The constants are what makes this kernel fat, and its waves share the SIMDs with other streams'
// kernels, so they are cut to 32 VGPRs: (i) input j + 8 of a 16-point DFT contributes w16^((j+8)k) = (-1)^k w16^(jk),
// and the rows of one MFMA tile are given outputs of ONE parity (row a of lane group ib is output 4 ib + a, resp.
// (a & 1) + 2 ib + 8 (a >> 1) in the DIT kernel), so K-chunk 1 reuses chunk 0's A registers: as they are for even
// rows, and for odd rows against the bytes XOR 0x7F (= 127 - x = -(x - 128) - 1: the negated operand, exact; the -1
// per entry is a row constant and sits in the C operand); (ii) the C operands are read from LDS next to each MFMA.
struct Frag {
  v4i a[8];
};
__device__ __forceinline__ void load_frag(Frag& f, const Tables* __restrict__ t) {
  const uint32_t lane = threadIdx.x & 63;
  const uint4* pa = reinterpret_cast<const uint4*>(t->a);
#pragma unroll
  for (int rb = 0; rb < 8; rb++) {
    const uint4 v = pa[rb * 64 + lane];
    f.a[rb] = (v4i){(int)v.x, (int)v.y, (int)v.z, (int)v.w};
  }
}
// the whole workgroup copies the C operands behind the data image (call before the first __syncthreads)
__device__ __forceinline__ void load_c_lds(uint32_t* __restrict__ c_lds, const Tables* __restrict__ t) {
  for (uint32_t i = threadIdx.x; i < (uint32_t)C_LDS_WORDS; i += blockDim.x) c_lds[i] = t->c[i];
}

// x[e]: input slot e of this lane (chunk e >> 1, element e & 1), any u64.  y[a]: output row a, reduced.
// cl: this lane group's C operands in LDS ([row block] stride 4 x v4i).
__device__ __forceinline__ void dft16(const uint64_t (&x)[4], const Frag& f, const v4i* __restrict__ cl, uint64_t (&y)[4]) {
  v4i b0, b1, b1n;
  b0[0] = (int)((uint32_t)x[0] ^ 0x80808080u); b0[1] = (int)((uint32_t)(x[0] >> 32) ^ 0x80808080u);
  b0[2] = (int)((uint32_t)x[1] ^ 0x80808080u); b0[3] = (int)((uint32_t)(x[1] >> 32) ^ 0x80808080u);
  b1[0] = (int)((uint32_t)x[2] ^ 0x80808080u); b1[1] = (int)((uint32_t)(x[2] >> 32) ^ 0x80808080u);
  b1[2] = (int)((uint32_t)x[3] ^ 0x80808080u); b1[3] = (int)((uint32_t)(x[3] >> 32) ^ 0x80808080u);
  b1n[0] = (int)((uint32_t)x[2] ^ 0x7F7F7F7Fu); b1n[1] = (int)((uint32_t)(x[2] >> 32) ^ 0x7F7F7F7Fu);
  b1n[2] = (int)((uint32_t)x[3] ^ 0x7F7F7F7Fu); b1n[3] = (int)((uint32_t)(x[3] >> 32) ^ 0x7F7F7F7Fu);
  uint64_t L[4], H[4];
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const v4i bb = (a & 1) ? b1n : b1;
    v4i dl = __builtin_amdgcn_mfma_i32_16x16x64_i8(f.a[2 * a], b0, cl[(2 * a) * 4], 0, 0, 0);
    dl = __builtin_amdgcn_mfma_i32_16x16x64_i8(f.a[2 * a], bb, dl, 0, 0, 0);
    v4i dh = __builtin_amdgcn_mfma_i32_16x16x64_i8(f.a[2 * a + 1], b0, cl[(2 * a + 1) * 4], 0, 0, 0);
    dh = __builtin_amdgcn_mfma_i32_16x16x64_i8(f.a[2 * a + 1], bb, dh, 0, 0, 0);
    L[a] = mxa::planes(dl);
    H[a] = mxa::planes(dh);
  }
  mxa::reduce_rows<4>(L, H, y);
}

#ifndef MXN_SETS_IN_FLIGHT
#define MXN_SETS_IN_FLIGHT 1   /* unroll factor of the set loops: 2 has more ILP and ~50 more VGPRs */
#endif
#define MXN_STR2(x) #x
#define MXN_STR(x) MXN_STR2(x)
#define MXN_PRAGMA_UNROLL_SETS _Pragma(MXN_STR(unroll MXN_SETS_IN_FLIGHT))
#ifndef MXN_WAVES
#define MXN_WAVES 3   /* waves per SIMD the register allocation aims at (128 VGPRs) */
#endif
// LDS addresses.  swz12 is linear over GF(2), and in every pass the set index m occupies position bits no other term
// touches, so address(m) = (lane constant) ^ swz12(m's bits): the lane constants are made once per pass.
__device__ __forceinline__ uint32_t mterm_hi(uint32_t m) { return (m << 8) ^ ((m & 2u) << 3); }   // swz12(m << 8)
__device__ __forceinline__ uint32_t mterm_lo(uint32_t m) { return (m << 4) ^ m; }                 // swz12(m << 4)
// DIF: input slot e (chunk e >> 1, element e & 1) is element j = 8c + 2kb + eps at field j; output row q is k = 4kb + q,
// stored at field bitrev4(k).  DIT: input slot e is element kin = 8c + 4eps + kb, found at field bitrev4(kin); output
// row q is j = (q & 1) + 2kb + 8(q >> 1), stored at field j.  (Chunk 1 = chunk 0's element + 8; a row's parity = q & 1.)
#define MXN_DIF_J(e) (8u * ((e) >> 1) + 2u * kb + ((e) & 1u))
#define MXN_DIF_K(q) (4u * kb + (q))
#define MXN_DIT_KIN(e) (8u * ((e) >> 1) + 4u * ((e) & 1u) + kb)
#define MXN_DIT_J(q) (((q) & 1u) + 2u * kb + 8u * ((q) >> 1))

// ---- decimation in frequency: natural -> bit-reversed.  X = log2(sub-blocks of 4096): block = 4096 << X points.
// 1-D grid of persistent workgroups: workgroup g takes units g, g + gridDim.x, ... (unit = column * blocks per column +
// block; a.n_units of them); 256 threads (X = 0) or 512 (two groups of 256).
template <int X>
__global__ void __launch_bounds__(X ? 512 : 256) __attribute__((amdgpu_waves_per_eu(MXN_WAVES, MXN_WAVES))) ntt_mx_dif_kernel(Ntt16Args a, const Tables* __restrict__ tab) {
  if (gridDim.x <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  extern __shared__ uint64_t buf[];
  constexpr int NSUB = 1 << X, NGRP = X ? 2 : 1, SPG = NSUB / NGRP;  // sub-blocks, thread groups, sub-blocks per group
  constexpr int MPT = 4 / NGRP;                                      // sets of the outermost pass per thread
  const uint32_t tid = threadIdx.x, ut = tid >> 8, t = tid & 255, lane = tid & 63, w = t >> 6, n = lane & 15,
                 kb = lane >> 4;
  Frag f;
  load_frag(f, tab);   // 8 x 16 bytes per lane: loaded once, the workgroup then walks over its share of the blocks
  uint32_t* c_lds = reinterpret_cast<uint32_t*>(buf + (4096u << X));
  load_c_lds(c_lds, tab);
  const v4i* cl = reinterpret_cast<const v4i*>(c_lds) + kb;   // [row block] stride 4: this lane group's C operands
  __syncthreads();
  const uint32_t log_bpc = a.log_n_total - (12 + X);  // blocks per column
#pragma unroll 1
  for (uint32_t unit = blockIdx.x; unit < a.n_units; unit += gridDim.x) {
    const uint64_t off = (uint64_t)(unit & ((1u << log_bpc) - 1)) << (12 + X);
    const uint64_t* src = a.in + (unit >> log_bpc) * a.in_stride + off;
    uint64_t* dst = a.out + (unit >> log_bpc) * a.out_stride + off;

    // outermost pass (S = 4096 inside every sub-block) straight from global memory, after the X radix-2 stages
    // that couple the sub-blocks
    {
      uint32_t wr[4];
#pragma unroll
      for (int q = 0; q < 4; q++) wr[q] = swz12((br4(MXN_DIF_K(q)) << 8) | (w << 6) | n);
MXN_PRAGMA_UNROLL_SETS
      for (int mi = 0; mi < MPT; mi++) {
        const uint32_t ms = ut * MPT + mi, G = 64 * w + 16 * ms + n, mt = mterm_lo(ms);
        uint64_t xin[NSUB][4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const uint32_t r = MXN_DIF_J(e) * 256 + G;
#pragma unroll
          for (int u = 0; u < NSUB; u++) xin[u][e] = src[u * 4096 + r];
          if constexpr (X > 0) {
            uint64_t yy[NSUB];
#pragma unroll
            for (int u = 0; u < NSUB; u++) yy[u] = xin[u][e];
            dif_butterflies<X>(yy, a.tw, r, 12, 0);
#pragma unroll
            for (int u = 0; u < NSUB; u++) xin[u][e] = yy[u];
          }
        }
#pragma unroll
        for (int u = 0; u < NSUB; u++) {
          uint64_t y[4], tw[4];
          dft16(xin[u], f, cl, y);
#pragma unroll
          for (int q = 0; q < 4; q++) tw[q] = tab->tw256[MXN_DIF_K(q) * 256 + G];
          gl::mul_n<4>(y, tw, y);
#pragma unroll
          for (int q = 0; q < 4; q++) buf[u * 4096 + (wr[q] ^ mt)] = y[q];
        }
      }
    }
    __syncthreads();
    // S = 256
    {
      uint32_t rd[4], wr[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        rd[e] = swz12((w << 10) | (MXN_DIF_J(e) << 4) | n);
        wr[e] = swz12((w << 10) | (br4(MXN_DIF_K(e)) << 4) | n);
      }
      uint64_t twv[4];
#pragma unroll
      for (int q = 0; q < 4; q++) twv[q] = tab->tw16[MXN_DIF_K(q) * 16 + n];
#pragma unroll 1
      for (int s = 0; s < SPG; s++) {
        uint64_t* sb = buf + (ut * SPG + s) * 4096;
MXN_PRAGMA_UNROLL_SETS
        for (int m = 0; m < 4; m++) {
          const uint32_t mt = mterm_hi(m);
          uint64_t x[4], y[4];
#pragma unroll
          for (int e = 0; e < 4; e++) x[e] = sb[rd[e] ^ mt];
          dft16(x, f, cl, y);
          gl::mul_n<4>(y, twv, y);
#pragma unroll
          for (int q = 0; q < 4; q++) sb[wr[q] ^ mt] = y[q];
        }
      }
    }
    __syncthreads();
    // S = 16 (no twiddle), then the 1/n of the inverse transform
    {
      uint32_t rd[4], wr[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        rd[e] = swz12((w << 10) | (n << 4) | MXN_DIF_J(e));
        wr[e] = swz12((w << 10) | (n << 4) | br4(MXN_DIF_K(e)));
      }
#pragma unroll 1
      for (int s = 0; s < SPG; s++) {
        uint64_t* sb = buf + (ut * SPG + s) * 4096;
MXN_PRAGMA_UNROLL_SETS
        for (int m = 0; m < 4; m++) {
          const uint32_t mt = mterm_hi(m);
          uint64_t x[4], y[4];
#pragma unroll
          for (int e = 0; e < 4; e++) x[e] = sb[rd[e] ^ mt];
          dft16(x, f, cl, y);
          if (a.out_scalar != 1) {
            const uint64_t sc[4] = {a.out_scalar, a.out_scalar, a.out_scalar, a.out_scalar};
            gl::mul_n<4>(y, sc, y);
          }
#pragma unroll
          for (int q = 0; q < 4; q++) sb[wr[q] ^ mt] = y[q];
        }
      }
    }
    __syncthreads();
    // coalesced store of the bit-reversed block
    {
      const uint32_t st0 = swz12(t);
#pragma unroll 1
      for (int s = 0; s < SPG; s++) {
        const uint32_t sub = (ut * SPG + s) * 4096;
#pragma unroll
        for (int i0 = 0; i0 < 16; i0 += 4) {
          uint64_t v[4];
#pragma unroll
          for (int i = 0; i < 4; i++) v[i] = buf[sub + (st0 ^ swz12((uint32_t)(i0 + i) << 8))];
          gl::canon_n<4>(v);
#pragma unroll
          for (int i = 0; i < 4; i++) dst[sub + (i0 + i) * 256 + t] = v[i];
        }
      }
    }
    __syncthreads();  // the LDS image is reused by the next block
  }
}

// ---- decimation in time: bit-reversed -> natural, optional per-coset input scale (the coset LDE).  One work item =
// one (column block, coset) in the XCD-aware id order of ntt16_dit_kernel (ntt.hip): id -> xcd = id % 8,
// k = id / 8, coset = k % n_cosets, unit = (k / n_cosets) * 8 + xcd = column * blocks_per_column + block.
// Input slot (chunk c, element eps) of lane kb is element kin = bitrev4(8c + 2kb + eps); output row a of lane ib is
// j = bitrev4(ib + 4a) (the matrix is built that way, see build_tables).
template <int X>
__global__ void __launch_bounds__(X ? 512 : 256) __attribute__((amdgpu_waves_per_eu(MXN_WAVES, MXN_WAVES))) ntt_mx_dit_kernel(Ntt16Args a, const Tables* __restrict__ tab) {
  if (gridDim.x <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  extern __shared__ uint64_t buf[];
  constexpr int NSUB = 1 << X, NGRP = X ? 2 : 1, SPG = NSUB / NGRP, MPT = 4 / NGRP;
  const uint32_t tid = threadIdx.x, ut = tid >> 8, t = tid & 255, lane = tid & 63, w = t >> 6, n = lane & 15,
                 kb = lane >> 4;
  Frag f;
  load_frag(f, tab);   // loaded once; the workgroup then walks over ids blockIdx.x, + gridDim.x (a multiple of 8), ...
  uint32_t* c_lds = reinterpret_cast<uint32_t*>(buf + (4096u << X));
  load_c_lds(c_lds, tab);
  const v4i* cl = reinterpret_cast<const v4i*>(c_lds) + kb;
  __syncthreads();
  const uint32_t log_bpc = a.log_n_total - (12 + X);
  const uint32_t n_ids = (a.n_units + 7) / 8 * 8 * a.n_cosets;
#pragma unroll 1
  for (uint32_t id = blockIdx.x; id < n_ids; id += gridDim.x) {
    const uint32_t kk = id >> 3;
    const uint32_t coset = kk % a.n_cosets, unit = (kk / a.n_cosets) * 8 + (id & 7);
    if (unit >= a.n_units) continue;  // padding of the last group of eight (whole workgroup skips together)
    const uint32_t col = unit >> log_bpc, blk = unit & ((1u << log_bpc) - 1);
    const uint64_t off = (uint64_t)blk << (12 + X);
    const uint64_t* src = a.in + col * a.in_stride + off;
    const uint64_t* sc = a.scale ? a.scale + ((uint64_t)coset << a.log_n_total) + off : nullptr;
    uint64_t* dst = a.out + col * a.out_stride + coset * a.out_coset_stride + off;

    // coalesced load (times the coset scale) into the swizzled LDS image
    {
      const uint32_t st0 = swz12(t);
#pragma unroll 1
      for (int s = 0; s < SPG; s++) {
        const uint32_t sub = (ut * SPG + s) * 4096;
#pragma unroll
        for (int i0 = 0; i0 < 16; i0 += 4) {
          uint64_t v[4];
#pragma unroll
          for (int i = 0; i < 4; i++) v[i] = src[sub + (i0 + i) * 256 + t];
          if (sc) {
            uint64_t s4[4];
#pragma unroll
            for (int i = 0; i < 4; i++) s4[i] = sc[sub + (i0 + i) * 256 + t];
            gl::mul_n<4>(v, s4, v);
          }
#pragma unroll
          for (int i = 0; i < 4; i++) buf[sub + (st0 ^ swz12((uint32_t)(i0 + i) << 8))] = v[i];
        }
      }
    }
    __syncthreads();
    // S = 16 (no twiddle): slot e sits at base + bitrev4(kin) = base + slot, row q goes to base + j
    {
      uint32_t rd[4], wr[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        rd[e] = swz12((w << 10) | (n << 4) | br4(MXN_DIT_KIN(e)));
        wr[e] = swz12((w << 10) | (n << 4) | MXN_DIT_J(e));
      }
#pragma unroll 1
      for (int s = 0; s < SPG; s++) {
        uint64_t* sb = buf + (ut * SPG + s) * 4096;
MXN_PRAGMA_UNROLL_SETS
        for (int m = 0; m < 4; m++) {
          const uint32_t mt = mterm_hi(m);
          uint64_t x[4], y[4];
#pragma unroll
          for (int e = 0; e < 4; e++) x[e] = sb[rd[e] ^ mt];
          dft16(x, f, cl, y);
#pragma unroll
          for (int q = 0; q < 4; q++) sb[wr[q] ^ mt] = y[q];
        }
      }
    }
    __syncthreads();
    // S = 256: twiddle w_256^(i kin) first
    {
      uint32_t rd[4], wr[4];
      uint64_t twv[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        rd[e] = swz12((w << 10) | (br4(MXN_DIT_KIN(e)) << 4) | n);
        wr[e] = swz12((w << 10) | (MXN_DIT_J(e) << 4) | n);
        twv[e] = tab->tw16[MXN_DIT_KIN(e) * 16 + n];
      }
#pragma unroll 1
      for (int s = 0; s < SPG; s++) {
        uint64_t* sb = buf + (ut * SPG + s) * 4096;
MXN_PRAGMA_UNROLL_SETS
        for (int m = 0; m < 4; m++) {
          const uint32_t mt = mterm_hi(m);
          uint64_t x[4], y[4];
#pragma unroll
          for (int e = 0; e < 4; e++) x[e] = sb[rd[e] ^ mt];
          gl::mul_n<4>(x, twv, x);
          dft16(x, f, cl, y);
#pragma unroll
          for (int q = 0; q < 4; q++) sb[wr[q] ^ mt] = y[q];
        }
      }
    }
    __syncthreads();
    // S = 4096 with the twiddle w_4096^(i kin) first, then the X radix-2 stages that join the sub-blocks, straight
    // to global memory
    {
      uint32_t rd[4];
#pragma unroll
      for (int e = 0; e < 4; e++) rd[e] = swz12((br4(MXN_DIT_KIN(e)) << 8) | (w << 6) | n);
MXN_PRAGMA_UNROLL_SETS
      for (int mi = 0; mi < MPT; mi++) {
        const uint32_t ms = ut * MPT + mi, G = 64 * w + 16 * ms + n, mt = mterm_lo(ms);
        uint64_t yo[NSUB][4], tw[4];
#pragma unroll
        for (int e = 0; e < 4; e++) tw[e] = tab->tw256[MXN_DIT_KIN(e) * 256 + G];
#pragma unroll
        for (int u = 0; u < NSUB; u++) {
          uint64_t x[4];
#pragma unroll
          for (int e = 0; e < 4; e++) x[e] = buf[u * 4096 + (rd[e] ^ mt)];
          gl::mul_n<4>(x, tw, x);
          dft16(x, f, cl, yo[u]);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t r = MXN_DIT_J(q) * 256 + G;
          if constexpr (X > 0) {
            uint64_t yy[NSUB];
#pragma unroll
            for (int u = 0; u < NSUB; u++) yy[u] = yo[u][q];
            dit_butterflies<X>(yy, a.tw, r, 12, 0);
#pragma unroll
            for (int u = 0; u < NSUB; u++) dst[u * 4096 + r] = gl::canon(yy[u]);
          } else {
            dst[r] = gl::canon(yo[0][q]);
          }
        }
      }
    }
    __syncthreads();  // the LDS image is reused by the next block
  }
}
#undef MXN_DIF_J
#undef MXN_DIF_K
#undef MXN_DIT_KIN
#undef MXN_DIT_J

// ---- host side: the constants of one (kind, direction) --------------------------------------------------------------
// kind 0 (DIF): column (c, kb, eps) is input j = 8c + 2kb + eps, row (ib, a) is output k = 4 ib + a.
// kind 1 (DIT): column is input kin = 8c + 4eps + kb, row is output j = (a & 1) + 2 ib + 8 (a >> 1).
// Only chunk c = 0 is stored: the coefficient of input index + 8 is (-1)^(output index) times it, and the parity of a
// row's output index is a & 1 in both kinds (dft16).
inline void digits8(uint64_t wv, int (&d)[8]) {  // balanced base-256 digits of wv or wv - p, whichever fits eight
  const unsigned __int128 lim = (unsigned __int128)127 * 0xFFFFFFFFFFFFFFFFULL / 255;
  __int128 v = wv <= (uint64_t)lim ? (__int128)wv : (__int128)wv - (__int128)gl::P;
  for (int q = 0; q < 8; q++) {
    int x = (int)(((v + 128) % 256 + 256) % 256) - 128;
    d[q] = x;
    v = (v - x) / 256;   // exact: v - x is a multiple of 256
  }
}
inline void build_tables(Tables& t, int kind, bool inverse) {
  uint64_t w4096 = gl::root(12);
  if (inverse) w4096 = gl::inv(w4096);
  const uint64_t w256 = gl::pow(w4096, 16), w16 = gl::pow(w4096, 256);
  auto in_index = [&](int kb, int eps) { return kind ? 4 * eps + kb : 2 * kb + eps; };             // chunk 0
  auto out_index = [&](int ib, int a) { return kind ? (a & 1) + 2 * ib + 8 * (a >> 1) : 4 * ib + a; };
  // bias: 2^22 + dd[q] per plane with sum_q (2^22 + dd[q]) 2^(8q) = 0 (mod p)
  unsigned __int128 base = 0;
  for (int q = 0; q < 8; q++) base += (unsigned __int128)(1u << 22) << (8 * q);
  const uint64_t delta = (uint64_t)(((unsigned __int128)gl::P - base % gl::P) % gl::P);
  static int dig[16][16][8][8];  // [out][in][p][q]
  for (int o = 0; o < 16; o++)
    for (int i = 0; i < 16; i++)
      for (int p = 0; p < 8; p++) {
        const uint64_t wv = gl::mulc(gl::pow(w16, (uint64_t)(o * i)), gl::canon((uint64_t)1 << (8 * p)));
        digits8(wv, dig[o][i][p]);
      }
  uint8_t* ab = reinterpret_cast<uint8_t*>(t.a);
  for (int rb = 0; rb < 8; rb++) {
    const int a = rb >> 1, h = rb & 1;
    for (int lane = 0; lane < 64; lane++) {
      const int r = lane & 15, kb = lane >> 4, o = out_index(r >> 2, a), q = 4 * h + (r & 3);
      for (int b = 0; b < 16; b++) ab[(rb * 64 + lane) * 16 + b] = (uint8_t)(int8_t)dig[o][in_index(kb, b >> 3)][b & 7][q];
    }
    for (int ib = 0; ib < 4; ib++)
      for (int reg = 0; reg < 4; reg++) {
        const int o = out_index(ib, a), q = 4 * h + reg;
        int s0 = 0;   // digit sum of the row over the 64 bytes of chunk 0 (chunk 1 multiplies the same A registers)
        for (int kb = 0; kb < 4; kb++)
          for (int eps = 0; eps < 2; eps++)
            for (int p = 0; p < 8; p++) s0 += dig[o][in_index(kb, eps)][p][q];
        // chunk 0 and an even row's chunk 1 see x - 128: + 128 s0 each; an odd row's chunk 1 sees 127 - x against
        // the un-negated digits: the true term is the computed one - 127 s0
        const int corr = 128 * s0 + ((a & 1) ? -127 * s0 : 128 * s0);
        t.c[(rb * 4 + ib) * 4 + reg] = (uint32_t)((1 << 22) + (int)((delta >> (8 * q)) & 0xFF) + corr);
      }
  }
  for (int k = 0; k < 16; k++) {
    for (int i = 0; i < 256; i++) t.tw256[k * 256 + i] = gl::pow(w4096, (uint64_t)(i * k));
    for (int i = 0; i < 16; i++) t.tw16[k * 16 + i] = gl::pow(w256, (uint64_t)(i * k));
  }
}

}  // namespace mxn
