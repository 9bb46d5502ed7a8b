// ntt.hip -- K2: batched Goldilocks NTT / iNTT / coset LDE over trace columns.
//
// Restates plonky2_field::fft (fft, ifft, coset_fft) and PolynomialBatch::from_values/from_coeffs
// (ifft -> zero-pad -> coset_fft(shift = 7)), reached from plonky_block_proof_gen/src/
// proof_gen.rs:44-52.  MI355X design (not upstream's recursive CPU FFT):
//   * values are natural order, coefficients are stored BIT-REVERSED, so the inverse transform is
//     a decimation-in-frequency pass (natural -> bit-reversed) and the forward one a
//     decimation-in-time pass (bit-reversed -> natural): no permutation pass ever touches HBM;
//   * the LDE on 7*<w_{n 2^r}> is 2^r independent n-point coset NTTs (the zero padding makes the
//     first r DIT stages trivial); coset t is written as one contiguous block ("coset-major");
//   * a column block of up to 2^14 elements (128 KiB) lives in LDS: one coalesced 8-byte-per-lane
//     read, log2(n) radix-2 butterfly stages done as radix-8 register passes (3 stages per LDS
//     round trip), one coalesced write: HBM traffic == algorithmic bytes (16*n per column);
//   * n > 2^14 adds strided global radix passes (each lane owns 2^k elements n/2^k apart, so every
//     wave access is a contiguous 512-byte run of one column).
// Roofline: HBM (8 TB/s peak).  Algorithmic bytes: NTT/iNTT 16*n*C; iNTT+LDE 8*n*C*(2+2^r).
#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <cstring>
#include <memory>
#include "common.hpp"
#include "gl.hpp"
#include "mx_arith.cuh"

namespace {

constexpr uint32_t LOG_BLK_MAX = 14;  // 2^14 * 8 B = 128 KiB of the 160 KiB LDS

// ---- radix-2^LOGR register passes.  x[m] sits at position base + m*sub of a block of size
// S = sub << LOGR; j = base mod sub.  `tw` is the table w_N^e, e < N/2, tw_shift = log2(N/S).
// The R/2 twiddle multiplies of a stage are independent: they go through gl::mul_n in groups of four
// (instruction-interleaved carry chains, gl.hpp), falling back to the one-at-a-time form for R < 8.
// J0: the caller guarantees j == 0 and log_sub == 0 (the innermost pass: 2^LOGR consecutive elements of a block
// that starts at a multiple of 2^LOGR).  The twiddle of a butterfly is then a compile-time power w^(m mod half),
// and a third to all of them are w^0 = 1: those multiplies are skipped (15 of the 32 of a radix-16 pass).
// The remaining ones are gathered into groups of four / three for gl::mul_n; after full unrolling every index
// below is a constant.
template <int N>
__device__ __forceinline__ void mul_group(uint64_t (&x)[16], const int (&idx)[4], const uint64_t (&d)[4], const uint64_t (&w)[4]) {
  if constexpr (N == 4) {
    uint64_t r[4];
    gl::mul_n<4>(d, w, r);
#pragma unroll
    for (int i = 0; i < 4; i++) x[idx[i]] = r[i];
  } else if constexpr (N == 3) {
    const uint64_t d3[3] = {d[0], d[1], d[2]}, w3[3] = {w[0], w[1], w[2]};
    uint64_t r[3];
    gl::mul_n<3>(d3, w3, r);
#pragma unroll
    for (int i = 0; i < 3; i++) x[idx[i]] = r[i];
  } else {
#pragma unroll
    for (int i = 0; i < N; i++) x[idx[i]] = gl::mul(d[i], w[i]);
  }
}

template <int LOGR, bool J0 = false>
__device__ __forceinline__ void dif_butterflies(uint64_t (&x)[1 << LOGR], const uint64_t* __restrict__ tw,
                                                uint32_t j, uint32_t log_sub, uint32_t tw_shift) {
  constexpr int R = 1 << LOGR, NB = R / 2;
#pragma unroll
  for (int s = 0; s < LOGR; s++) {
    const int half = R >> (s + 1);
    // butterfly k of the stage pairs x[m], x[m + half], m = (k / half) * 2 * half + k % half
    if constexpr (NB % 4 == 0 && J0 && LOGR == 4) {
      uint64_t dd[NB];
#pragma unroll
      for (int k0 = 0; k0 < NB; k0 += 4) {
        uint64_t a[4], b[4], d[4], r[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int k = k0 + i, m = (k / half) * 2 * half + (k % half);
          a[i] = x[m];
          b[i] = x[m + half];
        }
        gl::canon_n<4>(b);
        gl::add_n<4>(a, b, r);
        gl::sub_n<4>(a, b, d);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int k = k0 + i, m = (k / half) * 2 * half + (k % half);
          x[m] = r[i];
          dd[k] = d[i];
        }
      }
      // multiply only where the twiddle is not 1
      int cnt = 0;
      int idx[4] = {0, 0, 0, 0};
      uint64_t dv[4] = {0, 0, 0, 0}, wv[4] = {0, 0, 0, 0};
      uint64_t xx[16];
#pragma unroll
      for (int i = 0; i < 16; i++) xx[i] = x[i];
#pragma unroll
      for (int k = 0; k < NB; k++) {
        const int m = (k / half) * 2 * half + (k % half), e = m & (half - 1);
        if (e == 0) {
          xx[m + half] = dd[k];
        } else {
          idx[cnt] = m + half;
          dv[cnt] = dd[k];
          wv[cnt] = tw[(uint32_t)e << (tw_shift + s)];
          cnt++;
          if (cnt == 4) {
            mul_group<4>(xx, idx, dv, wv);
            cnt = 0;
          }
        }
      }
      if (cnt == 3) mul_group<3>(xx, idx, dv, wv);
      else if (cnt == 2) mul_group<2>(xx, idx, dv, wv);
      else if (cnt == 1) mul_group<1>(xx, idx, dv, wv);
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = xx[i];
    } else if constexpr (NB % 4 == 0) {
#pragma unroll
      for (int k0 = 0; k0 < NB; k0 += 4) {
        uint64_t a[4], b[4], d[4], w[4], r[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int k = k0 + i, m = (k / half) * 2 * half + (k % half);
          const uint32_t p = j + ((uint32_t)(m & (half - 1)) << log_sub);  // position in block of size S>>s
          w[i] = tw[p << (tw_shift + s)];
          a[i] = x[m];
          b[i] = x[m + half];
        }
        // lazily reduced values: only the operand that must be canonical is canonicalised
        gl::canon_n<4>(b);
        gl::add_n<4>(a, b, r);
        gl::sub_n<4>(a, b, d);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int k = k0 + i, m = (k / half) * 2 * half + (k % half);
          x[m] = r[i];
        }
        gl::mul_n<4>(d, w, r);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int k = k0 + i, m = (k / half) * 2 * half + (k % half);
          x[m + half] = r[i];
        }
        // groups in flight at once exhaust the SGPRs (each group of four holds ~9 carry masks) and the compiler
        // starts parking SGPRs in VGPR lanes; a v_writelane of a mask right after the asm that wrote it is a
        // hazard the recogniser cannot see, so the groups are kept apart instead
        if constexpr (LOGR >= 4) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int m = 0; m < R; m++) {
        if (m & half) continue;
        const uint64_t a = x[m], b = gl::canon(x[m + half]);
        x[m] = gl::add(a, b);
        if (J0 && (m & (half - 1)) == 0) {
          x[m + half] = gl::sub(a, b);  // twiddle w^0
        } else {
          const uint32_t p = j + ((uint32_t)(m & (half - 1)) << log_sub);
          x[m + half] = gl::mul(gl::sub(a, b), tw[p << (tw_shift + s)]);
        }
      }
    }
  }
}
template <int LOGR, bool J0 = false>
__device__ __forceinline__ void dit_butterflies(uint64_t (&x)[1 << LOGR], const uint64_t* __restrict__ tw,
                                                uint32_t j, uint32_t log_sub, uint32_t tw_shift) {
  constexpr int R = 1 << LOGR, NB = R / 2;
#pragma unroll
  for (int s = 0; s < LOGR; s++) {
    const int step = 1 << s;  // stage block size = sub << (s+1); tw_shift is for S = sub << LOGR
    if constexpr (NB % 4 == 0 && J0 && LOGR == 4) {
      // products first (only where the twiddle is not 1), then the additions of the whole stage
      uint64_t xx[16];
#pragma unroll
      for (int i = 0; i < 16; i++) xx[i] = x[i];
      int cnt = 0;
      int idx[4] = {0, 0, 0, 0};
      uint64_t dv[4] = {0, 0, 0, 0}, wv[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < NB; k++) {
        const int m = (k / step) * 2 * step + (k % step), e = m & (step - 1);
        if (e != 0) {
          idx[cnt] = m + step;
          dv[cnt] = x[m + step];
          wv[cnt] = tw[(uint32_t)e << (tw_shift + (LOGR - 1 - s))];
          cnt++;
          if (cnt == 4) {
            mul_group<4>(xx, idx, dv, wv);
            cnt = 0;
          }
        }
      }
      if (cnt == 3) mul_group<3>(xx, idx, dv, wv);
      else if (cnt == 2) mul_group<2>(xx, idx, dv, wv);
      else if (cnt == 1) mul_group<1>(xx, idx, dv, wv);
#pragma unroll
      for (int k0 = 0; k0 < NB; k0 += 4) {
        uint64_t a[4], r[4], hi[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int k = k0 + i, m = (k / step) * 2 * step + (k % step);
          a[i] = xx[m];
          r[i] = xx[m + step];
        }
        gl::canon_n<4>(r);
        gl::add_n<4>(a, r, hi);
        gl::sub_n<4>(a, r, lo);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int k = k0 + i, m = (k / step) * 2 * step + (k % step);
          x[m] = hi[i];
          x[m + step] = lo[i];
        }
      }
    } else if constexpr (NB % 4 == 0) {
#pragma unroll
      for (int k0 = 0; k0 < NB; k0 += 4) {
        uint64_t a[4], v[4], w[4], r[4], hi[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int k = k0 + i, m = (k / step) * 2 * step + (k % step);
          const uint32_t p = j + ((uint32_t)(m & (step - 1)) << log_sub);
          w[i] = tw[p << (tw_shift + (LOGR - 1 - s))];
          v[i] = x[m + step];
          a[i] = x[m];
        }
        gl::mul_n<4>(v, w, r);
        gl::canon_n<4>(r);
        gl::add_n<4>(a, r, hi);   // a: any u64, r: canonical -> lazily reduced results
        gl::sub_n<4>(a, r, lo);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int k = k0 + i, m = (k / step) * 2 * step + (k % step);
          x[m] = hi[i];
          x[m + step] = lo[i];
        }
        if constexpr (LOGR >= 4) __builtin_amdgcn_sched_barrier(0);  // see dif_butterflies
      }
    } else {
#pragma unroll
      for (int m = 0; m < R; m++) {
        if (m & step) continue;
        const uint32_t p = j + ((uint32_t)(m & (step - 1)) << log_sub);
        const uint64_t a = x[m];
        const uint64_t t = (J0 && (m & (step - 1)) == 0) ? gl::canon(x[m + step])
                                                         : gl::mulc(x[m + step], tw[p << (tw_shift + (LOGR - 1 - s))]);
        x[m] = gl::add(a, t);
        x[m + step] = gl::sub(a, t);
      }
    }
  }
}

// One pass over a block of 2^log_blk elements held in `buf` (LDS).  span_log = log2(S).
template <int LOGR, bool DIF>
__device__ __forceinline__ void lds_pass(uint64_t* buf, uint32_t log_blk, uint32_t span_log,
                                         const uint64_t* __restrict__ tw) {
  constexpr int R = 1 << LOGR;
  const uint32_t log_sub = span_log - LOGR, sub_mask = (1u << log_sub) - 1;
  const uint32_t n_q = 1u << (log_blk - LOGR);
  for (uint32_t q = threadIdx.x; q < n_q; q += blockDim.x) {
    const uint32_t j = q & sub_mask, blk = q >> log_sub;
    const uint32_t base = (blk << span_log) + j;
    uint64_t x[R];
#pragma unroll
    for (int m = 0; m < R; m++) x[m] = buf[base + ((uint32_t)m << log_sub)];
    if (DIF) dif_butterflies<LOGR>(x, tw, j, log_sub, log_blk - span_log);
    else dit_butterflies<LOGR>(x, tw, j, log_sub, log_blk - span_log);
#pragma unroll
    for (int m = 0; m < R; m++) buf[base + ((uint32_t)m << log_sub)] = x[m];
  }
}

struct LdsNttArgs {
  const uint64_t* in;
  uint64_t in_stride;        // elements between columns
  uint64_t* out;
  uint64_t out_stride;       // elements between columns
  uint64_t out_coset_stride; // elements between cosets (DIT LDE); 0 otherwise
  const uint64_t* tw;        // w_B^e (or inverse), e < B/2, B = 2^log_blk
  const uint64_t* scale;     // DIT: per-position input scale, [coset][n_total]; nullptr = none
  uint64_t out_scalar;       // DIF: multiply outputs (1/n); 1 = none
  uint32_t log_blk;          // block resident in LDS
  uint32_t log_n_total;      // whole column (>= log_blk)
};

// grid = (blocks per column, columns, cosets)
template <bool DIF>
__global__ void __launch_bounds__(1024) ntt_lds_kernel(LdsNttArgs a) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  extern __shared__ uint64_t buf[];
  const uint32_t n = 1u << a.log_blk;
  const uint64_t off = (uint64_t)blockIdx.x << a.log_blk;
  const uint64_t* src = a.in + blockIdx.y * a.in_stride + off;
  uint64_t* dst = a.out + blockIdx.y * a.out_stride + blockIdx.z * a.out_coset_stride + off;
  if (!DIF && a.scale) {
    const uint64_t* sc = a.scale + ((uint64_t)blockIdx.z << a.log_n_total) + off;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) buf[i] = gl::mulc(src[i], sc[i]);
  } else {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) buf[i] = src[i];
  }
  __syncthreads();
  const uint32_t rem = a.log_blk % 3;
  if (DIF) {
    uint32_t span = a.log_blk;
    for (; span >= 3 && span > rem; span -= 3) {
      lds_pass<3, true>(buf, a.log_blk, span, a.tw);
      __syncthreads();
    }
    if (rem == 1) lds_pass<1, true>(buf, a.log_blk, 1, a.tw);
    else if (rem == 2) lds_pass<2, true>(buf, a.log_blk, 2, a.tw);
    if (rem) __syncthreads();
  } else {
    if (rem == 1) lds_pass<1, false>(buf, a.log_blk, 1, a.tw);
    else if (rem == 2) lds_pass<2, false>(buf, a.log_blk, 2, a.tw);
    if (rem) __syncthreads();
    for (uint32_t span = rem + 3; span <= a.log_blk; span += 3) {
      lds_pass<3, false>(buf, a.log_blk, span, a.tw);
      __syncthreads();
    }
  }
  if (DIF && a.out_scalar != 1) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = gl::mulc(buf[i], a.out_scalar);
  } else {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = gl::canon(buf[i]);
  }
}

// ================================================================================================
// v2 LDS kernels for 2^12..2^14-point blocks: 16 elements per lane, four radix-2 stages per register
// pass, XOR-swizzled LDS image so every exchange is bank-conflict free (checked exhaustively, see
// DESIGN.md section 7), outer pass straight from/to global memory in coalesced 8-byte lanes, and
// for the LDE all 2^r cosets are produced from ONE read of the coefficients.
//   element index i (L bits).  Pass with register field [b+3:b]:  i = insert(t, m, b)
//   outermost field [L-1:L-4]:  i = m*T + t  (T = 2^L/16 lanes)  -> global position, coalesced
//   innermost field [3:0]:      i = bitrev(t)*16 + m             -> bit-reversed side, coalesced
// ================================================================================================
template <int L>
__device__ __forceinline__ uint32_t swz(uint32_t i) {
  uint32_t x;
  if (L == 14) x = (((i >> 6) & 7u) << 2) ^ ((i >> 9) & 31u);
  else if (L == 13) x = (((i >> 5) & 15u) << 1) ^ ((i >> 9) & 15u);
  else x = (((i >> 8) & 1u) << 4) ^ ((i >> 7) & 1u) ^ (((i >> 9) & 7u) << 1);
  return i ^ x;
}
__device__ __forceinline__ uint32_t insert4(uint32_t t, uint32_t m, uint32_t b) {
  return ((t >> b) << (b + 4)) | (m << b) | (t & ((1u << b) - 1));
}
// R-stage butterflies on the sub-arrays x[g + (u << SH)] (u = 0..2^R-1) for every g < 2^SH,
// or on x[(g << R) + u] when CONTIG.
template <int R, int SH, bool CONTIG, bool DIF>
__device__ __forceinline__ void sub_butterflies(uint64_t (&x)[16], const uint64_t* __restrict__ tw, uint32_t j,
                                                uint32_t j_step_log, uint32_t log_sub, uint32_t tw_shift) {
  constexpr int G = 16 >> R;
#pragma unroll
  for (int g = 0; g < G; g++) {
    uint64_t y[1 << R];
#pragma unroll
    for (int u = 0; u < (1 << R); u++) y[u] = CONTIG ? x[(g << R) + u] : x[g + (u << SH)];
    const uint32_t jj = CONTIG ? 0u : j + ((uint32_t)g << j_step_log);
    if (DIF) dif_butterflies<R, CONTIG>(y, tw, jj, log_sub, tw_shift);   // CONTIG is only used with j = 0, log_sub = 0
    else dit_butterflies<R, CONTIG>(y, tw, jj, log_sub, tw_shift);
#pragma unroll
    for (int u = 0; u < (1 << R); u++) {
      if (CONTIG) x[(g << R) + u] = y[u];
      else x[g + (u << SH)] = y[u];
    }
  }
}

struct Ntt16Args {
  const uint64_t* in;
  uint64_t in_stride;
  uint64_t* out;
  uint64_t out_stride, out_coset_stride;
  const uint64_t* tw;     // w_B^e, e < B/2, B = 2^L
  const uint64_t* scale;  // DIT: [coset][n_total] input scale, nullable
  uint64_t out_scalar;    // DIF: 1/n (1 = none)
  uint32_t log_n_total, n_cosets, n_units;  // n_units = columns * blocks per column (1-D grid mapping)
  const uint64_t* tw_top; // split kernels (F = 1): w_{2B}^e, e < B: the stage that joins the two halves
};

// ---- "split" forms (template parameter F = 1): a block of 2^(L+1) elements is transformed by TWO workgroups of
// the 2^L-point kernel, each doing half of the one radix-2 stage that couples the halves while it loads:
//   DIF: workgroup h reads x[i] and x[i + 2^L] and keeps (h = 0) x[i] + x[i + 2^L] or (h = 1) their twiddled
//        difference -- then it owns an independent 2^L-point transform (output half h);
//   DIT: the same split on the COEFFICIENT index (pairs of adjacent bit-reversed positions 2p, 2p+1), after which
//        workgroup h owns the outputs of parity h (stride-2 stores; the partner fills the other half of each line).
// Both read the whole 2^(L+1) block (the second read is an L2 hit: the pair sits on one XCD, see the grid
// mapping below) but no multiply is done twice.  Why: a 2^14-point block is ONE 128 KiB workgroup per CU, whose
// load, compute and store phases cannot overlap with anything (73 % VALU-busy); as two 64 KiB workgroups they
// overlap with each other (the 2^12 kernel reaches 88 %).  And a launch with few columns gets twice the
// workgroups, i.e. half the latency where the chip is not full.

// natural -> bit-reversed.  F = 0: grid = (blocks per column, columns).  F = 1: 1-D XCD-aware grid, id -> xcd = id % 8,
// k = id / 8, h = k % 2, unit = (k / 2) * 8 + xcd = column * (2^(L+1)-blocks per column) + block.
template <int L, int F>
__global__ void __launch_bounds__((1 << L) / 16) ntt16_dif_kernel(Ntt16Args a) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  extern __shared__ uint64_t buf[];
  constexpr uint32_t T = (1u << L) / 16;
  constexpr int RT = (L % 4 == 0) ? 4 : (L % 4);  // stages of the innermost pass
  const uint32_t t = threadIdx.x;
  const uint64_t* src;
  uint64_t* dst;
  uint64_t x[16];
  if (F) {
    const uint32_t id = blockIdx.x, k = id >> 3, h = k & 1, unit = (k >> 1) * 8 + (id & 7);
    if (unit >= a.n_units) return;  // padding of the last group of eight
    const uint32_t log_bpc = a.log_n_total - (L + 1);
    const uint32_t col = unit >> log_bpc, blk = unit & ((1u << log_bpc) - 1);
    const uint64_t off = (uint64_t)blk << (L + 1);
    src = a.in + col * a.in_stride + off;
    dst = a.out + col * a.out_stride + off + ((uint64_t)h << L);
#pragma unroll
    for (int m0 = 0; m0 < 16; m0 += 4) {
      uint64_t lo[4], hi[4], r[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        lo[i] = src[(m0 + i) * T + t];
        hi[i] = src[(m0 + i) * T + t + (1u << L)];
      }
      gl::canon_n<4>(hi);  // add_n / sub_n want b < p; caller memory may hold any u64 (bpg.h)
      if (h == 0) {
        gl::add_n<4>(lo, hi, r);
      } else {
        uint64_t d[4], w[4];
        gl::sub_n<4>(lo, hi, d);
#pragma unroll
        for (int i = 0; i < 4; i++) w[i] = a.tw_top[(m0 + i) * T + t];
        gl::mul_n<4>(d, w, r);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) x[m0 + i] = r[i];
    }
  } else {
    const uint64_t off = (uint64_t)blockIdx.x << L;
    src = a.in + blockIdx.y * a.in_stride + off;
    dst = a.out + blockIdx.y * a.out_stride + off;
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = src[m * T + t];
  }
  dif_butterflies<4>(x, a.tw, t, L - 4, 0);
#pragma unroll
  for (int m = 0; m < 16; m++) buf[swz<L>(m * T + t)] = x[m];
  __syncthreads();
#pragma unroll
  for (int b = L - 8; b > 0; b -= 4) {
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = buf[swz<L>(insert4(t, m, b))];
    dif_butterflies<4>(x, a.tw, t & ((1u << b) - 1), b, L - (b + 4));
#pragma unroll
    for (int m = 0; m < 16; m++) buf[swz<L>(insert4(t, m, b))] = x[m];
    __syncthreads();
  }
  const uint32_t base = gl::bitrev(t, L - 4) << 4;
#pragma unroll
  for (int m = 0; m < 16; m++) x[m] = buf[swz<L>(base | m)];
  sub_butterflies<RT, 0, true, true>(x, a.tw, 0, 0, 0, L - RT);
  // In-place DIF leaves position i holding coefficient bitrev(i): exactly the bit-reversed storage
  // order.  One more (conflict-free) LDS exchange turns "16 consecutive words per lane" into
  // coalesced 8-byte lanes for the global store.
#pragma unroll
  for (int m = 0; m < 16; m++) buf[swz<L>(base | m)] = x[m];
  __syncthreads();
  if (a.out_scalar != 1) {
#pragma unroll
    for (int m0 = 0; m0 < 16; m0 += 4) {
      uint64_t v[4], w[4], r[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        v[i] = buf[swz<L>((m0 + i) * T + t)];
        w[i] = a.out_scalar;
      }
      gl::mul_n<4>(v, w, r);
      gl::canon_n<4>(r);
#pragma unroll
      for (int i = 0; i < 4; i++) dst[(m0 + i) * T + t] = r[i];
    }
  } else {
#pragma unroll
    for (int m0 = 0; m0 < 16; m0 += 4) {
      uint64_t v[4];
#pragma unroll
      for (int i = 0; i < 4; i++) v[i] = buf[swz<L>((m0 + i) * T + t)];
      gl::canon_n<4>(v);
#pragma unroll
      for (int i = 0; i < 4; i++) dst[(m0 + i) * T + t] = v[i];
    }
  }
}

// bit-reversed -> natural, optional per-coset input scale.  One workgroup = one (column block, coset).
// XCD-aware 1-D grid: the 2^r coset workgroups of a block all re-read the same 8*2^L coefficient
// bytes, so they are given ids that agree modulo 8 and are adjacent in dispatch order: blocks are
// dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md, workgroup dispatch), which puts them on ONE
// XCD at about the same time and lets its L2 serve all but the first read (PMC: fetched bytes drop
// from 2^r x to ~1 x the coefficients).  id -> xcd = id % 8, k = id / 8, coset = k % n_cosets,
// unit = (k / n_cosets) * 8 + xcd, unit = column * blocks_per_column + block.  A speed choice only.
// F = 1 (split form, see Ntt16Args): k -> (coset, h) = ((k % (2 n_cosets)) / 2, k % 2), unit counts 2^(L+1)-blocks.
template <int L, int F>
__global__ void __launch_bounds__((1 << L) / 16) __attribute__((amdgpu_waves_per_eu(4, 8))) ntt16_dit_kernel(Ntt16Args a) {
  if (gridDim.x <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  extern __shared__ uint64_t buf[];
  constexpr uint32_t T = (1u << L) / 16;
  constexpr int RT = (L % 4 == 0) ? 4 : (L % 4);  // stages of the outermost pass
  const uint32_t id = blockIdx.x, k = id >> 3;
  const uint32_t per = a.n_cosets << F;
  const uint32_t ch = k % per, coset = ch >> F, h = F ? (ch & 1) : 0, unit = (k / per) * 8 + (id & 7);
  if (unit >= a.n_units) return;  // padding of the last group of eight (whole workgroup leaves together)
  const uint32_t log_bpc = a.log_n_total - (L + F);  // blocks per column
  const uint32_t col = unit >> log_bpc, blk = unit & ((1u << log_bpc) - 1);
  const uint32_t t = threadIdx.x;
  const uint64_t off = (uint64_t)blk << (L + F);
  const uint64_t* src = a.in + col * a.in_stride + off;
  const uint64_t* tw = a.tw;
  uint64_t x[16];
  const uint32_t base = gl::bitrev(t, L - 4) << 4;
  if (F) {
    // coefficient pairs (2p, 2p + 1): p = m*T + t is the input position of this workgroup's 2^L-point problem
    const ulonglong2* src2 = reinterpret_cast<const ulonglong2*>(src);
    const ulonglong2* sc2 = a.scale ? reinterpret_cast<const ulonglong2*>(a.scale + ((uint64_t)coset << a.log_n_total) + off) : nullptr;
    constexpr uint32_t BR4[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
#pragma unroll
    for (int m0 = 0; m0 < 16; m0 += 4) {
      uint64_t c0[4], c1[4], r[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const ulonglong2 c = src2[(m0 + i) * T + t];
        c0[i] = c.x;
        c1[i] = c.y;
      }
      if (sc2) {
        uint64_t s0[4], s1[4], p0[4], p1[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const ulonglong2 sc = sc2[(m0 + i) * T + t];
          s0[i] = sc.x;
          s1[i] = sc.y;
        }
        gl::mul_n<4>(c0, s0, p0);
        gl::mul_n<4>(c1, s1, p1);
        gl::canon_n<4>(p1);
#pragma unroll
        for (int i = 0; i < 4; i++) { c0[i] = p0[i]; c1[i] = p1[i]; }
      } else {
        gl::canon_n<4>(c1);  // unscaled: straight from caller memory, any u64
      }
      if (h == 0) {
        gl::add_n<4>(c0, c1, r);
      } else {
        uint64_t d[4], w[4];
        gl::sub_n<4>(c0, c1, d);
        // twiddle of coefficient index bitrev_L(p) = (bitrev_{L-4}(t) << 4) | bitrev_4(m)
#pragma unroll
        for (int i = 0; i < 4; i++) w[i] = a.tw_top[base | BR4[m0 + i]];
        gl::mul_n<4>(d, w, r);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) buf[swz<L>((m0 + i) * T + t)] = r[i];  // reduced, not canonical: fine as a butterfly input
    }
  } else if (a.scale) {
    const uint64_t* sc = a.scale + ((uint64_t)coset << a.log_n_total) + off;
#pragma unroll
    for (int m0 = 0; m0 < 16; m0 += 4) {
      uint64_t v[4], w[4], r[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        v[i] = src[(m0 + i) * T + t];
        w[i] = sc[(m0 + i) * T + t];
      }
      gl::mul_n<4>(v, w, r);
#pragma unroll
      for (int i = 0; i < 4; i++) buf[swz<L>((m0 + i) * T + t)] = r[i];  // reduced, not canonical: fine as a butterfly input
    }
  } else {
#pragma unroll
    for (int m = 0; m < 16; m++) buf[swz<L>(m * T + t)] = src[m * T + t];
  }
  __syncthreads();
  // innermost field [3:0]
#pragma unroll
  for (int m = 0; m < 16; m++) x[m] = buf[swz<L>(base | m)];
  dit_butterflies<4, true>(x, tw, 0, 0, L - 4);
#pragma unroll
  for (int m = 0; m < 16; m++) buf[swz<L>(base | m)] = x[m];
  __syncthreads();
#pragma unroll
  for (int b = 4; b + 4 <= L - RT; b += 4) {
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = buf[swz<L>(insert4(t, m, b))];
    dit_butterflies<4>(x, tw, t & ((1u << b) - 1), b, L - (b + 4));
#pragma unroll
    for (int m = 0; m < 16; m++) buf[swz<L>(insert4(t, m, b))] = x[m];
    __syncthreads();
  }
  // outermost field [L-1:L-4]: the top RT bits are still to do; lanes = low bits -> coalesced store.
  // These LDS addresses equal the ones of the first store; an opaque copy of t keeps the compiler from
  // carrying them across the whole body (it spills them to scratch = extra HBM writes).
  uint32_t t2 = t;
  asm volatile("" : "+v"(t2));
#pragma unroll
  for (int m = 0; m < 16; m++) x[m] = buf[swz<L>(m * T + t2)];
  sub_butterflies<RT, 4 - RT, false, false>(x, tw, t2, L - 4, L - RT, 0);
  uint64_t* dst = a.out + col * a.out_stride + coset * a.out_coset_stride + off;
#pragma unroll
  for (int m0 = 0; m0 < 16; m0 += 4) {
    uint64_t v[4] = {x[m0], x[m0 + 1], x[m0 + 2], x[m0 + 3]};
    gl::canon_n<4>(v);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (F) dst[2 * ((m0 + i) * T + t2) + h] = v[i];  // outputs of parity h
      else dst[(m0 + i) * T + t2] = v[i];
    }
  }
}

// The 2^14-point form of ntt16_dit_kernel<L, 0> as a PERSISTENT workgroup.  A 2^14-point block is 128 KiB of LDS: one
// workgroup per CU, so nothing overlaps its memory phases -- SQ counters (profiles/r5_ntt_stalls.txt): VALU busy
// 58..74 % of the launch, the rest is the coefficient load at the head of every block, the drain of its stores and the
// next workgroup's start.  Here the workgroup walks over its share of the (block, coset) items itself and loads the NEXT
// item's coefficients into registers while the last pass of the current one computes and stores (16 words per lane: the
// kernel has the registers, 48 of the 128 a four-wave SIMD allows).  The walk keeps id mod 8 (the XCD) and the
// coset-adjacent dispatch order of the one-shot grid: gridDim.x is a multiple of 8 * n_cosets.  No barrier between items:
// a lane's first LDS stores of an item go to the addresses its own last loads of the previous item read.
template <int L>
__global__ void __launch_bounds__((1 << L) / 16) __attribute__((amdgpu_waves_per_eu(4, 4)))
ntt16_dit_persist_kernel(Ntt16Args a, uint32_t n_items) {
  extern __shared__ uint64_t buf[];
  constexpr uint32_t T = (1u << L) / 16;
  constexpr int RT = (L % 4 == 0) ? 4 : (L % 4);
  const uint32_t t = threadIdx.x;
  const uint32_t log_bpc = a.log_n_total - L;
  const uint64_t* tw = a.tw;
  auto source_of = [&](uint32_t id) -> const uint64_t* {  // null: a padding id of the last group of eight
    const uint32_t k = id >> 3, unit = (k / a.n_cosets) * 8 + (id & 7);
    if (id >= n_items || unit >= a.n_units) return nullptr;
    const uint32_t col = unit >> log_bpc, blk = unit & ((1u << log_bpc) - 1);
    return a.in + col * a.in_stride + ((uint64_t)blk << L);
  };
  uint64_t xn[16];
  {
    const uint64_t* s0 = source_of(blockIdx.x);
    if (s0) {
#pragma unroll
      for (int m = 0; m < 16; m++) xn[m] = s0[m * T + t];
    }
  }
  for (uint32_t id = blockIdx.x; id < n_items; id += gridDim.x) {
    const uint32_t k = id >> 3, coset = k % a.n_cosets, unit = (k / a.n_cosets) * 8 + (id & 7);
    const uint64_t* next_src = source_of(id + gridDim.x);
    // opaque copies of the lane id, one per phase: every LDS and global address below is recomputed in the item it is
    // used in (hoisted out of the item loop they are ~100 registers: the kernel then spills, as its one-shot form would
    // without its own opaque copy)
    uint32_t ta = t, tb = t, tc = t;
    asm volatile("" : "+v"(ta));
    if (unit >= a.n_units) {  // padding (the whole workgroup agrees): nothing was prefetched for it; fetch for the next one
      if (next_src) {
#pragma unroll
        for (int m = 0; m < 16; m++) xn[m] = next_src[m * T + t];
      }
      continue;
    }
    const uint32_t col = unit >> log_bpc, blk = unit & ((1u << log_bpc) - 1);
    const uint64_t off = (uint64_t)blk << L;
    if (a.scale) {
      const uint64_t* sc = a.scale + ((uint64_t)coset << a.log_n_total) + off;
#pragma unroll
      for (int m0 = 0; m0 < 16; m0 += 4) {
        uint64_t v[4], w[4], r[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          v[i] = xn[m0 + i];
          w[i] = sc[(m0 + i) * T + ta];
        }
        gl::mul_n<4>(v, w, r);
#pragma unroll
        for (int i = 0; i < 4; i++) buf[swz<L>((m0 + i) * T + ta)] = r[i];
      }
    } else {
#pragma unroll
      for (int m = 0; m < 16; m++) buf[swz<L>(m * T + ta)] = xn[m];
    }
    __syncthreads();
    asm volatile("" : "+v"(tb));
    const uint32_t base = gl::bitrev(tb, L - 4) << 4;
    uint64_t x[16];
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = buf[swz<L>(base | m)];
    dit_butterflies<4, true>(x, tw, 0, 0, L - 4);
#pragma unroll
    for (int m = 0; m < 16; m++) buf[swz<L>(base | m)] = x[m];
    __syncthreads();
#pragma unroll
    for (int b = 4; b + 4 <= L - RT; b += 4) {
      asm volatile("" : "+v"(tc));
#pragma unroll
      for (int m = 0; m < 16; m++) x[m] = buf[swz<L>(insert4(tc, m, b))];
      dit_butterflies<4>(x, tw, tc & ((1u << b) - 1), b, L - (b + 4));
#pragma unroll
      for (int m = 0; m < 16; m++) buf[swz<L>(insert4(tc, m, b))] = x[m];
      __syncthreads();
    }
    // the next item's coefficients start their way now: they fly while the last pass computes and stores
    uint32_t t2 = t;
    asm volatile("" : "+v"(t2));
    if (next_src) {
#pragma unroll
      for (int m = 0; m < 16; m++) xn[m] = next_src[m * T + t2];
    }
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = buf[swz<L>(m * T + t2)];
    sub_butterflies<RT, 4 - RT, false, false>(x, tw, t2, L - 4, L - RT, 0);
    uint64_t* dst = a.out + col * a.out_stride + coset * a.out_coset_stride + off;
#pragma unroll
    for (int m0 = 0; m0 < 16; m0 += 4) {
      uint64_t v[4] = {x[m0], x[m0 + 1], x[m0 + 2], x[m0 + 3]};
      gl::canon_n<4>(v);
#pragma unroll
      for (int i = 0; i < 4; i++) dst[(m0 + i) * T + t2] = v[i];
    }
  }
}

#include "ntt_mx.cuh"  // the same blocks with the 16-point DFTs on the matrix cores

// Strided global pass for columns taller than one LDS block: the top LOGR stages (DIF) or the
// last LOGR stages (DIT) of the n-point transform.  Lane q owns elements q + m*(S>>LOGR).
// grid = (ceil(n >> LOGR / 256), columns, cosets).  `tw` is the n-point table.
template <int LOGR, bool DIF>
__global__ void __launch_bounds__(256)
ntt_global_pass_kernel(const uint64_t* in, uint64_t in_stride, uint64_t* out, uint64_t out_stride,
                       uint64_t coset_stride, uint32_t log_n, uint32_t span_log,
                       const uint64_t* __restrict__ tw) {
  constexpr int R = 1 << LOGR;
  const uint32_t log_sub = span_log - LOGR;
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (1u << (log_n - LOGR))) return;
  const uint32_t j = q & ((1u << log_sub) - 1), blk = q >> log_sub;
  const uint64_t base = ((uint64_t)blk << span_log) + j;
  const uint64_t* src = in + blockIdx.y * in_stride + blockIdx.z * coset_stride + base;
  uint64_t* dst = out + blockIdx.y * out_stride + blockIdx.z * coset_stride + base;
  uint64_t x[R];
#pragma unroll
  for (int m = 0; m < R; m++) x[m] = src[(uint64_t)m << log_sub];
  if (DIF) dif_butterflies<LOGR>(x, tw, j, log_sub, log_n - span_log);
  else dit_butterflies<LOGR>(x, tw, j, log_sub, log_n - span_log);
  if constexpr (R % 4 == 0) {
#pragma unroll
    for (int m0 = 0; m0 < R; m0 += 4) {
      uint64_t v[4] = {x[m0], x[m0 + 1], x[m0 + 2], x[m0 + 3]};
      gl::canon_n<4>(v);
#pragma unroll
      for (int i = 0; i < 4; i++) dst[(uint64_t)(m0 + i) << log_sub] = v[i];
    }
  } else {
#pragma unroll
    for (int m = 0; m < R; m++) dst[(uint64_t)m << log_sub] = gl::canon(x[m]);
  }
}


// Column pass THROUGH LDS for tall columns: the top A (6..8) stages of the DIF transform, or the last A stages of
// the DIT one, in one HBM round trip (the register pass above tops out at five stages: 32 elements per lane).
// The stage block (2^span_log elements) is viewed as R = 2^A rows of S = 2^(span_log - A) elements; a workgroup
// takes 16 adjacent elements of every row (128-byte segments: a wave touches four full cache lines per access)
// = R x 16 elements in LDS, and runs the R-point transform down each of its 16 columns as one radix-16 register
// pass over rows q + m*R/16 and one radix-2^(A-4) pass over the R/16 consecutive rows of a sub-block, with an
// LDS exchange between the two.  Row segments of a quarter wave are contiguous in LDS in both layouts.
// grid = (tiles per column = n / (R*16), columns, cosets), block = R threads.
template <int A, bool DIF>
__global__ void __launch_bounds__(1 << A)
ntt_tile_kernel(const uint64_t* in, uint64_t in_stride, uint64_t* out, uint64_t out_stride, uint64_t coset_stride,
                uint32_t log_n, uint32_t span_log, const uint64_t* __restrict__ tw) {
  constexpr int R = 1 << A, B = A - 4, R1 = 1 << B, G = 16 / R1;  // R1 rows per sub-block, G sub-blocks per lane
  __shared__ uint64_t tile[R * 16];
  const uint32_t log_s = span_log - A;                     // S = elements per row
  const uint32_t tiles_per_blk = 1u << (log_s - 4);
  const uint32_t blk = blockIdx.x / tiles_per_blk, s0 = (blockIdx.x % tiles_per_blk) << 4;
  const uint32_t s = threadIdx.x & 15, hi = threadIdx.x >> 4;  // hi: q in [0, R/16) for the radix-16 pass
  const uint64_t base = ((uint64_t)blk << span_log) + s0 + s;
  const uint64_t* src = in + blockIdx.y * in_stride + blockIdx.z * coset_stride + base;
  uint64_t* dst = out + blockIdx.y * out_stride + blockIdx.z * coset_stride + base;
  const uint32_t j16 = (hi << log_s) + s0 + s;             // position of row `hi` modulo the radix-16 stride
  uint64_t x[16];
  if (DIF) {
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = src[(uint64_t)(hi + m * (R / 16)) << log_s];
    dif_butterflies<4>(x, tw, j16, span_log - 4, log_n - span_log);
#pragma unroll
    for (int m = 0; m < 16; m++) tile[(hi + m * (R / 16)) * 16 + s] = x[m];
    __syncthreads();
    // sub-blocks hi*G + g, rows (hi*G + g)*R1 + u
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = tile[(hi * 16 + m) * 16 + s];
#pragma unroll
    for (int g = 0; g < G; g++) {
      uint64_t y[R1];
#pragma unroll
      for (int u = 0; u < R1; u++) y[u] = x[g * R1 + u];
      dif_butterflies<B>(y, tw, s0 + s, log_s, log_n - span_log + 4);
#pragma unroll
      for (int u = 0; u < R1; u++) x[g * R1 + u] = y[u];
    }
#pragma unroll
    for (int m0 = 0; m0 < 16; m0 += 4) {
      uint64_t v[4] = {x[m0], x[m0 + 1], x[m0 + 2], x[m0 + 3]};
      gl::canon_n<4>(v);
#pragma unroll
      for (int i = 0; i < 4; i++) dst[(uint64_t)(hi * 16 + m0 + i) << log_s] = v[i];
    }
  } else {
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = src[(uint64_t)(hi * 16 + m) << log_s];
#pragma unroll
    for (int g = 0; g < G; g++) {
      uint64_t y[R1];
#pragma unroll
      for (int u = 0; u < R1; u++) y[u] = x[g * R1 + u];
      dit_butterflies<B>(y, tw, s0 + s, log_s, log_n - span_log + 4);
#pragma unroll
      for (int u = 0; u < R1; u++) x[g * R1 + u] = y[u];
    }
#pragma unroll
    for (int m = 0; m < 16; m++) tile[(hi * 16 + m) * 16 + s] = x[m];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = tile[(hi + m * (R / 16)) * 16 + s];
    dit_butterflies<4>(x, tw, j16, span_log - 4, log_n - span_log);
#pragma unroll
    for (int m0 = 0; m0 < 16; m0 += 4) {
      uint64_t v[4] = {x[m0], x[m0 + 1], x[m0 + 2], x[m0 + 3]};
      gl::canon_n<4>(v);
#pragma unroll
      for (int i = 0; i < 4; i++) dst[(uint64_t)(hi + (m0 + i) * (R / 16)) << log_s] = v[i];
    }
  }
}

// table builders ------------------------------------------------------------------------------
__global__ void twiddle_table_kernel(uint64_t* out, uint32_t count, uint64_t w) {
  uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < count) out[e] = gl::pow(w, e);
}
// scale[t][pos] = (7 * w_M^t)^(+-bitrev_n(pos)),  M = n << rate_bits
__global__ void coset_scale_table_kernel(uint64_t* out, uint32_t log_n, uint32_t rate_bits, uint64_t w_m,
                                         int inverse) {
  uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= (1u << log_n)) return;
  uint32_t t = blockIdx.y;
  uint64_t g = gl::mulc(gl::GENERATOR, gl::pow(w_m, t));
  if (inverse) g = gl::inv(g);
  out[((uint64_t)t << log_n) + pos] = gl::pow(g, gl::bitrev(pos, log_n));
}
__global__ void bitrev_permute_kernel(uint64_t* cols, uint64_t stride, uint32_t log_n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1u << log_n)) return;
  uint32_t j = gl::bitrev(i, log_n);
  if (i < j) {
    uint64_t* c = cols + blockIdx.y * stride;
    uint64_t a = c[i], b = c[j];
    c[i] = b;
    c[j] = a;
  }
}

// ---- per-device table cache ----------------------------------------------------------------
struct TableKey {
  int dev, kind;  // kind 0: fwd twiddles, 1: inv twiddles, 2: coset scales, 3: inverse coset scales
  uint32_t log_n, rate_bits;
  bool operator<(const TableKey& o) const {
    return std::tie(dev, kind, log_n, rate_bits) < std::tie(o.dev, o.kind, o.log_n, o.rate_bits);
  }
};
std::mutex g_table_mu;
std::map<TableKey, uint64_t*> g_tables;

}  // namespace

namespace bpg {

// Tables are built once per (device, size) on the null stream and kept for the process lifetime.
int get_table(int kind, uint32_t log_n, uint32_t rate_bits, const uint64_t** out) {
  int dev = 0;
  BPG_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_table_mu);
  TableKey key{dev, kind, log_n, kind >= 2 ? rate_bits : 0};
  auto it = g_tables.find(key);
  if (it != g_tables.end()) {
    *out = it->second;
    return BP_OK;
  }
  uint64_t* d = nullptr;
  if (kind >= 2) {
    uint64_t count = (uint64_t)1 << (log_n + rate_bits);
    BPG_HIP(hipMalloc(&d, count * 8));
    dim3 grid(ceil_div((uint64_t)1 << log_n, 256), 1u << rate_bits);
    coset_scale_table_kernel<<<grid, 256, 0, 0>>>(d, log_n, rate_bits, gl::root(log_n + rate_bits), kind == 3);
  } else {
    uint32_t count = log_n ? (1u << (log_n - 1)) : 1;
    BPG_HIP(hipMalloc(&d, (uint64_t)count * 8));
    uint64_t w = gl::root(log_n);
    if (kind == 1) w = gl::inv(w);
    twiddle_table_kernel<<<ceil_div(count, 256), 256, 0, 0>>>(d, count, w);
  }
  BPG_LAUNCH_CHECK();
  BPG_HIP(hipStreamSynchronize(0));
  g_tables[key] = d;
  *out = d;
  return BP_OK;
}

// Matrix-core form of the block kernels (ntt_mx.cuh).  0: never; 1: 2^12- and 2^13-point blocks; 2: 2^14-point blocks
// too (tests); 3 (default): 2^13-point blocks -- where it wins a little -- while the device is not loaded (fewer than
// six provers at work).  Measured (bench.py --ntt-mx 1 / 0 back to back on one box; HISTORY.md, round 2):
//   first version, MFMA constants in 96 VGPRs, 224-240 VGPRs = two waves per SIMD (profiles/r2_ntt_mx_probe.txt,
//   r2_ntt_mx_block_ab.txt): alone level at 2^12 points, +7 % LDE / +17 % inverse at 2^13 x 135 rate 8, -25 % at 2^14
//   (spills); under the 24-stream block run 29.8 against 34.1 txn-proofs/s -- its fat waves crowd out the Poseidon
//   kernels' waves;
//   this version, constants cut to 32 VGPRs (chunk 1 = +-chunk 0, C operands in LDS), 136-168 VGPRs = three waves per
//   SIMD (profiles/r2_ntt_mx_lean_ab.txt): alone level at 2^12, +3..4 % at 2^13 x 135, still behind at 2^14; under
//   load 34.7 against 35.3.
// It issues half the VALU instructions of the butterfly kernels, but every 16 elements of a pass cost one MFMA, which
// blocks its SIMD for ~12 cycles (tools/mfma_probe.hip), and the kernel stays latency-bound at three waves: no win
// over the VALU kernels on this workload, so it is used only where it measures ahead.
bool device_loaded();  // hash_kernels.hip
static std::atomic<int> g_ntt_mx{3};
static bool use_ntt_mx(uint32_t log_blk, bool dit) {
  const int m = g_ntt_mx.load(std::memory_order_relaxed);
  if (m == 3) return log_blk == 13 && !device_loaded();
  if (m == 4) return !dit && log_blk <= 13;   // measurement modes: one direction only
  if (m == 5) return dit && log_blk <= 13;
  return m == 2 || (m == 1 && log_blk <= 13);
}
// Constants of the matrix-core kernels, one device image per (device, kind, direction), built once.
static std::mutex g_mx_mu;
static std::map<std::tuple<int, int, int>, mxn::Tables*> g_mx_tabs;
// Persistent grid of the matrix-core kernels: every workgroup loads the 24 KB of MFMA constants once and then walks
// over its share of the work items, so the grid is what is resident at once (a multiple of 8: the XCD-aware id
// mapping of the DIT kernel needs id mod 8 to stay fixed along a workgroup's walk).
static std::atomic<int> g_mx_wg_per_cu{0};  // 0 = default (2 for 2^12-point blocks, 1 above)
static uint32_t mx_grid(uint32_t items, uint32_t log_blk) {
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  int per = g_mx_wg_per_cu.load(std::memory_order_relaxed);
  if (per <= 0) per = log_blk == 12 ? 2 : 1;
  const uint32_t resident = ((uint32_t)cus * per + 7) / 8 * 8;
  return items < resident ? items : resident;
}
static int get_mx_tables(int kind, bool inverse, const mxn::Tables** out) {
  int dev = 0;
  BPG_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mx_mu);
  const auto key = std::make_tuple(dev, kind, inverse ? 1 : 0);
  auto it = g_mx_tabs.find(key);
  if (it != g_mx_tabs.end()) {
    *out = it->second;
    return BP_OK;
  }
  auto host = std::make_unique<mxn::Tables>();
  mxn::build_tables(*host, kind, inverse);
  mxn::Tables* d = nullptr;
  BPG_HIP(hipMalloc(&d, sizeof(mxn::Tables)));
  BPG_HIP(hipMemcpy(d, host.get(), sizeof(mxn::Tables), hipMemcpyHostToDevice));
  g_mx_tabs[key] = d;
  *out = d;
  return BP_OK;
}

// Plan for a column of 2^log_n elements: the LDS-resident block size and the global passes that come before it
// (DIF) or after it (DIT).  Columns up to 2^14 are one block.  The 2^12-point LDS kernel is the most efficient one
// (four workgroups per CU: load / compute / store of different workgroups overlap), so taller columns use the
// smallest block that ONE global pass allows: a register pass of up to four stages (log_n <= 17; a fifth stage =
// 32 elements and 31 scalar row offsets per lane does not fit the SGPR file next to the carry masks), the LDS tile
// pass of six to eight stages (log_n <= 22); beyond that register passes come on top (2^23..2^26: three round
// trips, 2^27..2^30: four).
struct NttPlan {
  uint32_t log_blk;      // LDS block
  uint32_t tile_bits;    // stages of the tile pass (0 = none), acting on span log_blk + tile_bits
  uint32_t n_reg;        // register passes above that
  uint32_t reg_bits[4];  // from the outermost span (log_n) inwards
};
static NttPlan plan_ntt(uint32_t log_n) {
  NttPlan p{};
  if (log_n <= LOG_BLK_MAX) { p.log_blk = log_n; return p; }
  if (log_n <= 17) { p.log_blk = log_n <= 16 ? 12 : 13; p.n_reg = 1; p.reg_bits[0] = log_n - p.log_blk; return p; }
  if (log_n <= 20) { p.log_blk = 12; p.tile_bits = log_n - 12; return p; }
  if (log_n <= 22) { p.log_blk = log_n - 8; p.tile_bits = 8; return p; }
  p.log_blk = 14; p.tile_bits = 8;
  uint32_t left = log_n - 22;
  p.n_reg = (left + 3) / 4;
  for (uint32_t i = 0; i < p.n_reg; i++) {
    const uint32_t k = (left + (p.n_reg - i) - 1) / (p.n_reg - i);
    p.reg_bits[i] = k;
    left -= k;
  }
  return p;
}
static uint32_t pick_log_blk(uint32_t log_n) { return plan_ntt(log_n).log_blk; }

// Persistent 2^14-point DIT workgroups (ntt16_dit_persist_kernel): 0 = the one-shot grid (default), 1 = on; resident
// workgroups: one per CU.  Measured and NOT adopted (profiles/r5_ntt_stalls.txt): 2^14 x 2432 at rate 2 takes 893..919 us
// persistent against 795..857 us one-shot on the same box -- the prefetched words push the kernel to the 128-VGPR edge
// of four waves per SIMD (140 bytes of scratch per lane), and a one-shot grid already overlaps a finishing workgroup's
// store drain with its successor's start on the same CU.
static std::atomic<int> g_ntt_persist{0};
static std::atomic<int> g_ntt_persist_wgs{256};
// Split form (Ntt16Args), out of place only.
static std::atomic<int> g_ntt_split{0};  // 0 = automatic, 1 = never, 2 = wherever possible, 3 = automatic + small DIT launches (measurement knobs)
static bool use_split(uint32_t log_blk, uint64_t workgroups, const void* src, const void* dst, bool dit) {
  const int mode = g_ntt_split.load(std::memory_order_relaxed);
  // never in place: each of the two workgroups reads the WHOLE block while its partner may already be storing
  if (mode == 1 || log_blk < 13 || src == dst) return false;
  if (mode == 2) return true;
  // The DIT form stays off by default: its stride-2 stores (each workgroup writes every other word of a line)
  // inflate the HBM write traffic of the 2^14 x 2432 LDE by 1.66x (PMC WRITE_SIZE: 1,036,585 KiB against
  // 622,592 KiB algorithmic) for a 6 % shorter launch, and it gains nothing on small launches.  The DIF form
  // writes contiguous halves: 2^14-point blocks always (+23 %), 2^13-point blocks when the launch has too few
  // workgroups to fill the chip (2^13 x 16: 23 -> 19 us).
  if (dit) return mode == 3 && log_blk == 13 && workgroups < 512;  // 3: measurement mode, small DIT launches only
  return log_blk == 14 || workgroups < 512;
}

static uint32_t lds_threads(uint32_t log_blk) {
  uint32_t t = log_blk >= 3 ? (1u << (log_blk - 3)) : 1;
  return t < 64 ? 64 : (t > 1024 ? 1024 : t);
}

template <bool DIF>
static int launch_global_passes(const uint64_t* in, uint64_t in_stride, uint64_t* out, uint64_t out_stride,
                                uint64_t coset_stride, uint32_t n_cols, uint32_t n_cosets, uint32_t log_n,
                                uint32_t log_blk, const uint64_t* tw, hipStream_t st) {
  // DIF: spans log_n, log_n-k1, ... down to > log_blk;  DIT: the same passes in reverse order.
  const NttPlan p = plan_ntt(log_n);
  uint32_t spans[8], bits[8], tiled[8], cnt = 0;
  uint32_t span = log_n;
  for (uint32_t i = 0; i < p.n_reg; i++) { spans[cnt] = span; bits[cnt] = p.reg_bits[i]; tiled[cnt] = 0; cnt++; span -= p.reg_bits[i]; }
  if (p.tile_bits) { spans[cnt] = span; bits[cnt] = p.tile_bits; tiled[cnt] = 1; cnt++; span -= p.tile_bits; }
  if (span != log_blk || log_blk != p.log_blk) return fail(BP_ERR_DEVICE, "NTT plan mismatch (log_n=%u)", log_n);
  for (uint32_t idx = 0; idx < cnt; idx++) {
    const uint32_t i = DIF ? idx : cnt - 1 - idx;
    const uint64_t* src = (idx == 0) ? in : out;
    const uint64_t src_stride = (idx == 0) ? in_stride : out_stride;
    if (tiled[i]) {
      const dim3 grid(1u << (log_n - bits[i] - 4), n_cols, n_cosets);
      switch (bits[i]) {
        case 6: ntt_tile_kernel<6, DIF><<<grid, 64, 0, st>>>(src, src_stride, out, out_stride, coset_stride, log_n, spans[i], tw); break;
        case 7: ntt_tile_kernel<7, DIF><<<grid, 128, 0, st>>>(src, src_stride, out, out_stride, coset_stride, log_n, spans[i], tw); break;
        default: ntt_tile_kernel<8, DIF><<<grid, 256, 0, st>>>(src, src_stride, out, out_stride, coset_stride, log_n, spans[i], tw); break;
      }
    } else {
      const dim3 grid(ceil_div((uint64_t)1 << (log_n - bits[i]), 256), n_cols, n_cosets);
      switch (bits[i]) {
        case 1: ntt_global_pass_kernel<1, DIF><<<grid, 256, 0, st>>>(src, src_stride, out, out_stride, coset_stride, log_n, spans[i], tw); break;
        case 2: ntt_global_pass_kernel<2, DIF><<<grid, 256, 0, st>>>(src, src_stride, out, out_stride, coset_stride, log_n, spans[i], tw); break;
        case 3: ntt_global_pass_kernel<3, DIF><<<grid, 256, 0, st>>>(src, src_stride, out, out_stride, coset_stride, log_n, spans[i], tw); break;
        case 4: ntt_global_pass_kernel<4, DIF><<<grid, 256, 0, st>>>(src, src_stride, out, out_stride, coset_stride, log_n, spans[i], tw); break;
        default: return fail(BP_ERR_DEVICE, "NTT plan asks for a %u-stage register pass", bits[i]);
      }
    }
    BPG_LAUNCH_CHECK();
  }
  return BP_OK;
}

// values (natural) -> coefficients (bit-reversed), scaled by 1/n.  in may equal out.
int intt_nat2br(const uint64_t* in, uint64_t in_stride, uint64_t* out, uint64_t out_stride, uint32_t log_n,
                uint32_t n_cols, bool inverse, hipStream_t st) {
  if (n_cols == 0) return BP_OK;
  const uint32_t log_blk = pick_log_blk(log_n);
  const uint64_t *tw_n = nullptr, *tw_b = nullptr;
  int rc;
  if ((rc = get_table(inverse ? 1 : 0, log_blk, 0, &tw_b))) return rc;
  const uint64_t* src = in;
  uint64_t src_stride = in_stride;
  if (log_n > LOG_BLK_MAX) {
    if ((rc = get_table(inverse ? 1 : 0, log_n, 0, &tw_n))) return rc;
    if ((rc = launch_global_passes<true>(in, in_stride, out, out_stride, 0, n_cols, 1, log_n, log_blk, tw_n, st))) return rc;
    src = out;
    src_stride = out_stride;
  }
  if (log_blk >= 12) {
    Ntt16Args b{};
    b.in = src; b.in_stride = src_stride; b.out = out; b.out_stride = out_stride; b.out_coset_stride = 0;
    b.tw = tw_b; b.scale = nullptr; b.out_scalar = inverse ? gl::inv((uint64_t)1 << log_n) : 1;
    b.log_n_total = log_n; b.n_cosets = 1; b.n_units = 0;
    KernelTimer kt(PROF_INTT_DIF, st, 16.0 * (double)n_cols * (double)((uint64_t)1 << log_n));
    if (use_ntt_mx(log_blk, false)) {
      const mxn::Tables* tab = nullptr;
      if ((rc = get_mx_tables(0, inverse, &tab))) return rc;
      b.n_units = n_cols << (log_n - log_blk);
      const uint32_t g = mx_grid(b.n_units, log_blk);
      if (log_blk == 12) mxn::ntt_mx_dif_kernel<0><<<g, 256, (8u << 12) + 4 * mxn::C_LDS_WORDS, st>>>(b, tab);
      else if (log_blk == 13) mxn::ntt_mx_dif_kernel<1><<<g, 512, (8u << 13) + 4 * mxn::C_LDS_WORDS, st>>>(b, tab);
      else mxn::ntt_mx_dif_kernel<2><<<g, 512, (8u << 14) + 4 * mxn::C_LDS_WORDS, st>>>(b, tab);
    } else if (use_split(log_blk, (uint64_t)n_cols << (log_n - log_blk), src, out, false)) {
      // two workgroups of the next smaller kernel per block (see Ntt16Args)
      if ((rc = get_table(inverse ? 1 : 0, log_blk - 1, 0, &b.tw))) return rc;
      b.tw_top = tw_b;
      b.n_units = n_cols << (log_n - log_blk);
      const dim3 grid1((b.n_units + 7) / 8 * 8 * 2);
      if (log_blk == 14) ntt16_dif_kernel<13, 1><<<grid1, 512, 8u << 13, st>>>(b);
      else ntt16_dif_kernel<12, 1><<<grid1, 256, 8u << 12, st>>>(b);
    } else {
      const dim3 grid16(1u << (log_n - log_blk), n_cols);
      if (log_blk == 12) ntt16_dif_kernel<12, 0><<<grid16, 256, 8u << 12, st>>>(b);
      else if (log_blk == 13) ntt16_dif_kernel<13, 0><<<grid16, 512, 8u << 13, st>>>(b);
      else ntt16_dif_kernel<14, 0><<<grid16, 1024, 8u << 14, st>>>(b);
    }
    BPG_LAUNCH_CHECK();
    return BP_OK;
  }
  LdsNttArgs a{};
  a.in = src; a.in_stride = src_stride; a.out = out; a.out_stride = out_stride; a.out_coset_stride = 0;
  a.tw = tw_b; a.scale = nullptr;
  a.out_scalar = inverse ? gl::inv((uint64_t)1 << log_n) : 1;
  a.log_blk = log_blk; a.log_n_total = log_n;
  dim3 grid(1u << (log_n - log_blk), n_cols, 1);
  size_t lds = (size_t)8 << log_blk;
  {
    KernelTimer kt(PROF_INTT_DIF, st, 16.0 * (double)n_cols * (double)((uint64_t)1 << log_n));
    ntt_lds_kernel<true><<<grid, lds_threads(log_blk), lds, st>>>(a);
  }
  BPG_LAUNCH_CHECK();
  return BP_OK;
}

// coefficients (bit-reversed) -> values (natural) on n_cosets cosets; scale (nullable) is the
// [coset][n] input scale table.  Coset t goes to out + t*coset_stride.
int ntt_br2nat(const uint64_t* in, uint64_t in_stride, uint64_t* out, uint64_t out_stride, uint64_t coset_stride,
               uint32_t log_n, uint32_t n_cols, uint32_t n_cosets, const uint64_t* scale, bool inverse,
               hipStream_t st) {
  if (n_cols == 0) return BP_OK;
  const uint32_t log_blk = pick_log_blk(log_n);
  const uint64_t *tw_n = nullptr, *tw_b = nullptr;
  int rc;
  if ((rc = get_table(inverse ? 1 : 0, log_blk, 0, &tw_b))) return rc;
  if (log_blk >= 12) {
    Ntt16Args b{};
    b.in = in; b.in_stride = in_stride; b.out = out; b.out_stride = out_stride; b.out_coset_stride = coset_stride;
    b.tw = tw_b; b.scale = scale; b.out_scalar = 1; b.log_n_total = log_n; b.n_cosets = n_cosets;
    b.n_units = n_cols << (log_n - log_blk);
    {
      // (attached: the kernel's own start / stop events -- this family is bench.py's `roofline`)
      KernelTimer kt(PROF_LDE_DIT, st, 8.0 * (double)n_cols * (double)((uint64_t)1 << log_n) * (1.0 + n_cosets), true);
      if (use_ntt_mx(log_blk, true)) {
        const mxn::Tables* tab = nullptr;
        if ((rc = get_mx_tables(1, inverse, &tab))) return rc;
        const uint32_t g = mx_grid((b.n_units + 7) / 8 * 8 * n_cosets, log_blk);
        if (log_blk == 12) BPG_LAUNCH_TIMED(kt, mxn::ntt_mx_dit_kernel<0>, g, 256, (8u << 12) + 4 * mxn::C_LDS_WORDS, st, b, tab);
        else if (log_blk == 13) BPG_LAUNCH_TIMED(kt, mxn::ntt_mx_dit_kernel<1>, g, 512, (8u << 13) + 4 * mxn::C_LDS_WORDS, st, b, tab);
        else BPG_LAUNCH_TIMED(kt, mxn::ntt_mx_dit_kernel<2>, g, 512, (8u << 14) + 4 * mxn::C_LDS_WORDS, st, b, tab);
      } else if (use_split(log_blk, (uint64_t)b.n_units * n_cosets, in, out, true)) {
        if ((rc = get_table(inverse ? 1 : 0, log_blk - 1, 0, &b.tw))) return rc;
        b.tw_top = tw_b;
        const dim3 grid1((b.n_units + 7) / 8 * 8 * n_cosets * 2);
        if (log_blk == 14) BPG_LAUNCH_TIMED(kt, HIP_KERNEL_NAME(ntt16_dit_kernel<13, 1>), grid1, 512, 8u << 13, st, b);
        else BPG_LAUNCH_TIMED(kt, HIP_KERNEL_NAME(ntt16_dit_kernel<12, 1>), grid1, 256, 8u << 12, st, b);
      } else {
        const dim3 grid16((b.n_units + 7) / 8 * 8 * n_cosets);
        if (log_blk == 12) BPG_LAUNCH_TIMED(kt, HIP_KERNEL_NAME(ntt16_dit_kernel<12, 0>), grid16, 256, 8u << 12, st, b);
        else if (log_blk == 13) BPG_LAUNCH_TIMED(kt, HIP_KERNEL_NAME(ntt16_dit_kernel<13, 0>), grid16, 512, 8u << 13, st, b);
        else {
          // 2^14-point blocks: one workgroup per CU either way; with more items than CUs the workgroups are persistent and
          // prefetch (ntt16_dit_persist_kernel).  Grid: a multiple of 8 * n_cosets, at most one workgroup per CU.
          const uint32_t items = grid16.x, group = 8 * n_cosets;
          const uint32_t resident = std::max<uint32_t>(group, (uint32_t)g_ntt_persist_wgs.load(std::memory_order_relaxed) / group * group);
          if (g_ntt_persist.load(std::memory_order_relaxed) && items > resident && in != out)
            BPG_LAUNCH_TIMED(kt, HIP_KERNEL_NAME(ntt16_dit_persist_kernel<14>), dim3(resident), 1024, 8u << 14, st, b, items);
          else
            BPG_LAUNCH_TIMED(kt, HIP_KERNEL_NAME(ntt16_dit_kernel<14, 0>), grid16, 1024, 8u << 14, st, b);
        }
      }
    }
    BPG_LAUNCH_CHECK();
  } else {
  LdsNttArgs a{};
  a.in = in; a.in_stride = in_stride; a.out = out; a.out_stride = out_stride; a.out_coset_stride = coset_stride;
  a.tw = tw_b; a.scale = scale; a.out_scalar = 1; a.log_blk = log_blk; a.log_n_total = log_n;
  dim3 grid(1u << (log_n - log_blk), n_cols, n_cosets);
  size_t lds = (size_t)8 << log_blk;
  {
    // algorithmic bytes: coefficients read once, every coset written once
    KernelTimer kt(PROF_LDE_DIT, st, 8.0 * (double)n_cols * (double)((uint64_t)1 << log_n) * (1.0 + n_cosets), true);
    BPG_LAUNCH_TIMED(kt, ntt_lds_kernel<false>, grid, lds_threads(log_blk), lds, st, a);
  }
  BPG_LAUNCH_CHECK();
  }
  if (log_n > LOG_BLK_MAX) {
    if ((rc = get_table(inverse ? 1 : 0, log_n, 0, &tw_n))) return rc;
    if ((rc = launch_global_passes<false>(out, out_stride, out, out_stride, coset_stride, n_cols, n_cosets, log_n,
                                          log_blk, tw_n, st)))
      return rc;
  }
  return BP_OK;
}

int bitrev_permute(uint64_t* cols, uint64_t stride, uint32_t log_n, uint32_t n_cols, hipStream_t st) {
  dim3 grid(ceil_div((uint64_t)1 << log_n, 256), n_cols);
  bitrev_permute_kernel<<<grid, 256, 0, st>>>(cols, stride, log_n);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}

static int init_ntt_kernels_once() {
  // allow the full 128 KiB dynamic LDS block
  BPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_lds_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 8 << LOG_BLK_MAX));
  BPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_lds_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 8 << LOG_BLK_MAX));
  BPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt16_dif_kernel<14, 0>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 8 << 14));
  BPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt16_dit_kernel<14, 0>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 8 << 14));
  BPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt16_dit_persist_kernel<14>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 8 << 14));
  BPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt16_dif_kernel<13, 0>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 8 << 13));
  BPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt16_dit_kernel<13, 0>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 8 << 13));
  BPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mxn::ntt_mx_dif_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (8 << 13) + 4 * mxn::C_LDS_WORDS));
  BPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mxn::ntt_mx_dif_kernel<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (8 << 14) + 4 * mxn::C_LDS_WORDS));
  BPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mxn::ntt_mx_dit_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (8 << 13) + 4 * mxn::C_LDS_WORDS));
  BPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mxn::ntt_mx_dit_kernel<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (8 << 14) + 4 * mxn::C_LDS_WORDS));
  BPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt16_dif_kernel<13, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 8 << 13));
  BPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt16_dit_kernel<13, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 8 << 13));
  return BP_OK;
}
int init_ntt_kernels() {
  static std::once_flag once;
  static int rc = BP_OK;
  std::call_once(once, [] { rc = init_ntt_kernels_once(); });
  return rc;
}

}  // namespace bpg

extern "C" {

void bp_tune_ntt_split(int mode) { bpg::g_ntt_split.store(mode); }
void bp_tune_ntt_persist(int on, int resident_workgroups) {
  bpg::g_ntt_persist.store(on != 0);
  if (resident_workgroups > 0) bpg::g_ntt_persist_wgs.store(resident_workgroups);
}
void bp_tune_ntt_mx(int mode) { bpg::g_ntt_mx.store(mode < 0 || mode > 5 ? 3 : mode); }
void bp_tune_ntt_mx_wg_per_cu(int n) { bpg::g_mx_wg_per_cu.store(n); }

// Host only (no device call): the constants of the matrix-core NTT kernels as the device gets them, for the CPU tests
// that pin them to the integer model (tools/ntt_mx_model.py).  out_a: 8192 bytes, out_c: 128 i32,
// out_tw256: 4096 u64, out_tw16: 256 u64.
int bp_debug_ntt_mx_tables(int kind, int inverse, uint8_t* out_a, int32_t* out_c, uint64_t* out_tw256,
                           uint64_t* out_tw16) try {
  if ((kind != 0 && kind != 1) || !out_a || !out_c || !out_tw256 || !out_tw16)
    return bpg::fail(BP_ERR_INVALID_INPUT, "bp_debug_ntt_mx_tables: bad argument");
  auto t = std::make_unique<mxn::Tables>();
  {
    std::lock_guard<std::mutex> lk(bpg::g_mx_mu);  // build_tables keeps its digit scratch in a static array
    mxn::build_tables(*t, kind, inverse != 0);
  }
  memcpy(out_a, t->a, sizeof(t->a));
  memcpy(out_c, t->c, sizeof(t->c));
  memcpy(out_tw256, t->tw256, sizeof(t->tw256));
  memcpy(out_tw16, t->tw16, sizeof(t->tw16));
  return BP_OK;
}
BPG_ABI_CATCH("bp_debug_ntt_mx_tables")

int bp_ntt_batch(uint64_t* d_cols, uint32_t log_n, uint32_t n_cols, uint64_t col_stride, int dir, void* stream) try {
  if (n_cols == 0) return BP_OK;
  if (!d_cols) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_ntt_batch: null buffer");
  if (log_n > 30 || col_stride < ((uint64_t)1 << log_n))
    return bpg::fail(BP_ERR_INVALID_INPUT, "bp_ntt_batch: bad shape (log_n=%u stride=%llu)", log_n,
                     (unsigned long long)col_stride);
  hipStream_t st = bpg::as_stream(stream);
  int rc;
  if ((rc = bpg::init_ntt_kernels())) return rc;
  switch (dir) {
    case BP_NTT_FWD_BR2NAT:
      return bpg::ntt_br2nat(d_cols, col_stride, d_cols, col_stride, 0, log_n, n_cols, 1, nullptr, false, st);
    case BP_NTT_INV_NAT2BR:
      return bpg::intt_nat2br(d_cols, col_stride, d_cols, col_stride, log_n, n_cols, true, st);
    case BP_NTT_FWD_NAT:
      if ((rc = bpg::bitrev_permute(d_cols, col_stride, log_n, n_cols, st))) return rc;
      return bpg::ntt_br2nat(d_cols, col_stride, d_cols, col_stride, 0, log_n, n_cols, 1, nullptr, false, st);
    case BP_NTT_INV_NAT:
      if ((rc = bpg::intt_nat2br(d_cols, col_stride, d_cols, col_stride, log_n, n_cols, true, st))) return rc;
      return bpg::bitrev_permute(d_cols, col_stride, log_n, n_cols, st);
    default:
      return bpg::fail(BP_ERR_INVALID_INPUT, "bp_ntt_batch: unknown dir %d", dir);
  }
}
BPG_ABI_CATCH("bp_ntt_batch")

// The first half of PolynomialBatch::from_values as its own entry: values stay, coefficients go to a second
// buffer.  Out of place lets a 2^13 / 2^14-point block be transformed by two workgroups (Ntt16Args).
int bp_intt_batch(const uint64_t* d_values, uint64_t in_stride, uint64_t* d_coeffs_out, uint64_t out_stride,
                  uint32_t log_n, uint32_t n_cols, void* stream) try {
  if (n_cols == 0) return BP_OK;
  if (!d_values || !d_coeffs_out) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_intt_batch: null buffer");
  const uint64_t n = (uint64_t)1 << log_n;
  if (log_n > 30 || in_stride < n || out_stride < n) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_intt_batch: bad shape (log_n=%u)", log_n);
  const uint64_t span_in = in_stride * (n_cols - 1) + n, span_out = out_stride * (n_cols - 1) + n;
  if (d_values != d_coeffs_out && d_values < d_coeffs_out + span_out && d_coeffs_out < d_values + span_in)
    return bpg::fail(BP_ERR_INVALID_INPUT, "bp_intt_batch: buffers overlap (use the same pointer for in place)");
  int rc;
  if ((rc = bpg::init_ntt_kernels())) return rc;
  return bpg::intt_nat2br(d_values, in_stride, d_coeffs_out, out_stride, log_n, n_cols, true, bpg::as_stream(stream));
}
BPG_ABI_CATCH("bp_intt_batch")

int bp_lde_batch(const uint64_t* d_in, uint64_t in_stride, uint64_t* d_coeffs_out, uint64_t coeffs_stride,
                 uint64_t* d_lde_out, uint64_t lde_stride, uint32_t log_n, uint32_t rate_bits, uint32_t n_cols,
                 int from_coeffs, void* stream) try {
  if (n_cols == 0) return BP_OK;
  const uint64_t n = (uint64_t)1 << log_n;
  if (!d_in || !d_lde_out || (!from_coeffs && !d_coeffs_out))
    return bpg::fail(BP_ERR_INVALID_INPUT, "bp_lde_batch: null buffer");
  if (log_n + rate_bits > 30 || rate_bits > 4 || in_stride < n || lde_stride < (n << rate_bits) ||
      (d_coeffs_out && coeffs_stride < n))
    return bpg::fail(BP_ERR_INVALID_INPUT, "bp_lde_batch: bad shape (log_n=%u rate_bits=%u)", log_n, rate_bits);
  if (d_coeffs_out && d_coeffs_out != d_in) {  // as bp_intt_batch: the split kernels read a block while its partner stores
    const uint64_t span_in = in_stride * (n_cols - 1) + n, span_c = coeffs_stride * (n_cols - 1) + n;
    if (d_in < d_coeffs_out + span_c && d_coeffs_out < d_in + span_in)
      return bpg::fail(BP_ERR_INVALID_INPUT, "bp_lde_batch: input and coefficient buffers overlap (use the same pointer for in place)");
  }
  {
    const uint64_t* src = from_coeffs ? d_in : d_coeffs_out;  // what the forward transform reads
    const uint64_t span_s = (from_coeffs ? in_stride : coeffs_stride) * (n_cols - 1) + n;
    const uint64_t span_l = lde_stride * (n_cols - 1) + (n << rate_bits);
    if (src < d_lde_out + span_l && d_lde_out < src + span_s)
      return bpg::fail(BP_ERR_INVALID_INPUT, "bp_lde_batch: the LDE output overlaps the coefficients it is computed from");
  }
  hipStream_t st = bpg::as_stream(stream);
  int rc;
  if ((rc = bpg::init_ntt_kernels())) return rc;
  const uint64_t* coeffs = d_in;
  uint64_t cstride = in_stride;
  if (!from_coeffs) {
    if ((rc = bpg::intt_nat2br(d_in, in_stride, d_coeffs_out, coeffs_stride, log_n, n_cols, true, st))) return rc;
    coeffs = d_coeffs_out;
    cstride = coeffs_stride;
  } else if (d_coeffs_out && d_coeffs_out != d_in) {
    BPG_HIP(hipMemcpy2DAsync(d_coeffs_out, coeffs_stride * 8, d_in, in_stride * 8, n * 8, n_cols,
                             hipMemcpyDeviceToDevice, st));
  }
  const uint64_t* scale = nullptr;
  if ((rc = bpg::get_table(2, log_n, rate_bits, &scale))) return rc;
  return bpg::ntt_br2nat(coeffs, cstride, d_lde_out, lde_stride, n, log_n, n_cols, 1u << rate_bits, scale, false,
                         st);
}
BPG_ABI_CATCH("bp_lde_batch")

}  // extern "C"
