// gl.hpp -- Goldilocks field (p = 2^64 - 2^32 + 1) and its quadratic extension for gfx950 device
// code and the C++ host side of the prover (challenger, verifier).
//
// Field / hash / extension degree are the ones fixed by the reference's type aliases:
// plonky_block_proof_gen/src/types.rs:10-18 (GoldilocksField, PoseidonGoldilocksConfig, D = 2).
// Representation: every value stored to memory is canonical (< p).  In registers a value is any
// u64 congruent to the element ("reduced"), and each helper states what it needs and returns.
//
// CDNA4 notes: there is no 64x64 multiplier; a 64x64->128 product is four v_mad_u64_u32 (full rate:
// 4.7 cycles per wave64 like every other VOP3, tools/issue_rate.hip) and the Goldilocks reduction uses 2^64 = 2^32 - 1, 2^96 = -1.  The compiler's
// rendering of mul_wide + reduce128 is 25 VALU instructions; the device forms below (namespace cc,
// mul, mul_n, DotAcc) keep the carries as explicit SGPR masks and need 15.
// The field arithmetic itself is VALU work; the matrix cores are used where a constant matrix acts on many elements
// (the Poseidon MDS layer, poseidon_mx.cuh; opt-in the NTT's 16-point DFTs, ntt_mx.cuh) through byte planes.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

#define GL_HD __host__ __device__ __forceinline__

namespace gl {

constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
constexpr uint64_t EPS = 0xFFFFFFFFULL;  // 2^64 mod p
constexpr uint64_t GENERATOR = 7;        // multiplicative generator = LDE coset shift
constexpr uint64_t TWO_ADIC_ROOT = 1753635133440165772ULL;  // 7^((p-1)/2^32)
constexpr uint64_t W = 7;                // extension non-residue: X^2 = 7

#if defined(__HIP__)  // both passes of a HIP compilation parse these; only the device pass emits them
// ---- carry-chain primitives (device only).  The compiler never uses the carry-out of
// v_mad_u64_u32 and forms (x, 0) register pairs with v_mov for every 32->64-bit addend (64-bit VGPR
// operands must be even-aligned), which makes its modular multiply 25 VALU instructions.  With the
// carries kept as explicit wave masks in SGPR pairs the same product + reduction is 15.  Each
// one-instruction asm statement that CONSUMES a carry carries its own `s_nop 1`: on gfx940+ a VALU
// write of an SGPR needs 2 wait states before a VALU read of it, and the compiler's hazard
// recogniser does not look inside inline asm.  (Other waves of the SIMD issue during the nop.)
// The same blindness holds the other way round: a compiler-generated DPP / permlane read of a VGPR
// that an asm statement has just written is not given its 2 wait states either, so code that feeds
// asm results straight into cross-lane moves inserts them itself (poseidon.cuh, permute_quad).
namespace cc {
typedef uint64_t mask;  // wave-wide carry mask, lives in an SGPR pair
__device__ __forceinline__ uint64_t mad_co(uint32_t a, uint32_t b, uint64_t c, mask& co) {
  uint64_t d;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(co) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ uint64_t mad_eps_co(uint32_t a, uint64_t c, mask& co) {  // a*(2^32-1) + c
  uint64_t d;
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(d), "=s"(co) : "v"(a), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t add_co(uint32_t a, uint32_t b, mask& co) {
  uint32_t d;
  asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(d), "=s"(co) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t addc_co(uint32_t a, uint32_t b, mask ci, mask& co) {  // a + b + ci
  uint32_t d;
  asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(d), "=s"(co) : "v"(a), "v"(b), "s"(ci));
  return d;
}
__device__ __forceinline__ uint32_t addc0_co(uint32_t a, mask ci, mask& co) {  // a + ci
  uint32_t d;
  asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(d), "=s"(co) : "v"(a), "s"(ci));
  return d;
}
__device__ __forceinline__ uint32_t sub_co(uint32_t a, uint32_t b, mask& co) {
  uint32_t d;
  asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(d), "=s"(co) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t subb_co(uint32_t a, uint32_t b, mask ci, mask& co) {  // a - b - ci
  uint32_t d;
  asm("s_nop 1\n\tv_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(d), "=s"(co) : "v"(a), "v"(b), "s"(ci));
  return d;
}
__device__ __forceinline__ uint32_t subb0_co(uint32_t a, mask ci, mask& co) {  // a - ci
  uint32_t d;
  asm("s_nop 1\n\tv_subbrev_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(d), "=s"(co) : "v"(a), "s"(ci));
  return d;
}
__device__ __forceinline__ uint32_t sel_eps(mask m) {  // m ? 2^32-1 : 0
  uint32_t d;
  asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(d) : "s"(m));
  return d;
}
__device__ __forceinline__ uint64_t mk64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
#include "gl_cc.inc"  // the same instructions in groups of N = 3, 4 independent elements (no s_nop needed)
}  // namespace cc
#endif
// canon / add / sub / addc / subc keep their portable form (the compiler's compare-and-select code): single-element
// carry-chain versions of them were measured SLOWER in the latency-bound STARK kernels (fri_fold 13.3 -> 17.2 us,
// power_vector 9.8 -> 11.5 us: every mask consumer needs its own s_nop), although a microbenchmark of the
// compiler's pattern looks worse (tools/issue_rate5.hip).  The interleaved group forms add_n / sub_n / canon_n
// below have no nops and are what the NTT butterflies use (+5 %).
GL_HD uint64_t canon(uint64_t a) {
  return a >= P ? a - P : a;
}

// a: any u64, b: canonical.  Result: reduced (any u64).
GL_HD uint64_t add(uint64_t a, uint64_t b) {
  uint64_t s = a + b;
  return s < b ? s + EPS : s;  // on wrap: s < b < p so s + EPS cannot wrap again
}
// a: any u64, b: canonical.  Result: reduced.
GL_HD uint64_t sub(uint64_t a, uint64_t b) {
  uint64_t d = a - b;
  return a < b ? d - EPS : d;  // on borrow: d = a + 2^64 - b > EPS
}
// both canonical -> canonical
GL_HD uint64_t addc(uint64_t a, uint64_t b) {
  uint64_t s = a + b;
  return (s < a || s >= P) ? s - P : s;
}
GL_HD uint64_t subc(uint64_t a, uint64_t b) {
  return a >= b ? a - b : a + (P - b);
}
GL_HD uint64_t negc(uint64_t a) {
  return a ? P - a : 0;
}

// 128-bit (lo, hi) -> reduced u64
GL_HD uint64_t reduce128(uint64_t lo, uint64_t hi) {
  uint64_t hh = hi >> 32, hl = hi & EPS;
  uint64_t t0 = lo - hh;
  if (lo < hh) t0 -= EPS;
  uint64_t t1 = (hl << 32) - hl;  // hl * (2^32 - 1)
  uint64_t r = t0 + t1;
  return r < t1 ? r + EPS : r;
}

GL_HD void mul_wide(uint64_t a, uint64_t b, uint64_t& lo, uint64_t& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  // four 32x32->64 multiply-adds (v_mad_u64_u32)
  uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
  uint64_t p00 = (uint64_t)a0 * b0;
  uint64_t mid = (uint64_t)a0 * b1 + (p00 >> 32);
  uint64_t mid2 = (uint64_t)a1 * b0 + (uint32_t)mid;
  hi = (uint64_t)a1 * b1 + (mid >> 32) + (mid2 >> 32);
  lo = (mid2 << 32) | (uint32_t)p00;
#else
  unsigned __int128 x = (unsigned __int128)a * b;
  lo = (uint64_t)x;
  hi = (uint64_t)(x >> 64);
#endif
}

// any x any -> reduced
GL_HD uint64_t mul(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  // a*b = P + M*2^32 + C*2^96 + Q*2^64 with P = a0*b0, M + C*2^64 = a0*b1 + a1*b0, Q = a1*b1.
  // In 32-bit limbs and with 2^64 = EPS, 2^96 = -1 (mod p):
  //   a*b = U + S*EPS - (Q1 + C + c2),  U = (P0, P1 + M0 mod 2^32),  S + c2*2^32 = M1 + Q0 + carry(P1 + M0)
  const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
  cc::mask C, c1, c2, c3, cx, bw, bw2, b3;
  const uint64_t P = (uint64_t)a0 * b0;
  const uint64_t M = cc::mad_co(a1, b0, (uint64_t)a0 * b1, C);
  const uint64_t Q = (uint64_t)a1 * b1;  // Q1 <= 2^32 - 2, so Q1 + C fits
  const uint32_t p1 = cc::add_co((uint32_t)(P >> 32), (uint32_t)M, c1);
  const uint32_t S = cc::addc_co((uint32_t)(M >> 32), (uint32_t)Q, c1, c2);
  const uint32_t K = cc::addc0_co((uint32_t)(Q >> 32), C, cx);
  // V = U + S*EPS mod 2^64 (carry c3: + EPS, cannot wrap), then - K - c2 (borrow bw2: - EPS, cannot borrow);
  // +-EPS under a mask is two instructions (see mul_n)
  const uint64_t V = cc::mad_eps_co(S, cc::mk64((uint32_t)P, p1), c3);
  const uint32_t t0 = cc::subb0_co((uint32_t)V, c3, b3);
  const uint32_t t1 = cc::addc0_co((uint32_t)(V >> 32), c3 & ~b3, cx);
  const uint32_t u0 = cc::subb_co(t0, K, c2, bw);
  const uint32_t u1 = cc::subb0_co(t1, bw, bw2);
  const uint32_t r0 = cc::addc0_co(u0, bw2, b3);
  const uint32_t r1 = cc::subb0_co(u1, bw2 & ~b3, cx);
  return cc::mk64(r0, r1);
#else
  uint64_t lo, hi;
  mul_wide(a, b, lo, hi);
  return reduce128(lo, hi);
#endif
}
#if defined(__HIP__)
// N (= 3 or 4) independent products at once, instruction-interleaved: same arithmetic as mul(), but
// every carry consumer sits N-1 >= 2 instructions behind its producer, so no s_nop is spent.  This is
// the form the throughput kernels use (12 S-boxes of a Poseidon round, 8 butterflies of an NTT stage).
template <int N>
__device__ __forceinline__ void mul_n(const uint64_t (&a)[N], const uint64_t (&b)[N], uint64_t (&r)[N]) {
  static_assert(N == 3 || N == 4, "groups of 3 or 4");
  // every step overwrites one of its operands (gl_cc.inc): per element P, M, Q and one scratch word
  uint32_t a1[N], b0[N], t0[N], t1[N], m0[N], m1[N], q0[N], q1[N];
  uint64_t P[N], M[N], Q[N];
  cc::mask C[N], c1[N], c2[N], c3[N], bw[N], bw2[N], b3[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    const uint32_t a0 = (uint32_t)a[i], b1 = (uint32_t)(b[i] >> 32);
    a1[i] = (uint32_t)(a[i] >> 32);
    b0[i] = (uint32_t)b[i];
    P[i] = (uint64_t)a0 * b0[i];
    M[i] = (uint64_t)a0 * b1;
    Q[i] = (uint64_t)a1[i] * b1;  // Q1 <= 2^32 - 2, so Q1 + C fits
  }
  cc::mad_co(M, C, a1, b0);  // M += a1*b0, carry C (weight 2^96 = -1)
#pragma unroll
  for (int i = 0; i < N; i++) {
    t0[i] = (uint32_t)P[i]; t1[i] = (uint32_t)(P[i] >> 32); m0[i] = (uint32_t)M[i]; m1[i] = (uint32_t)(M[i] >> 32);
    q0[i] = (uint32_t)Q[i]; q1[i] = (uint32_t)(Q[i] >> 32);
  }
  cc::add_co(t1, c1, m0);        // U = (P0, P1 + M0)
  cc::addc_co(m1, c2, q0, c1);   // S = M1 + Q0 + c1 (mod 2^32), carry c2
  cc::addc0_cv(q1, C);           // K = Q1 + C (no carry: Q1 <= 2^32 - 2)
#pragma unroll
  for (int i = 0; i < N; i++) P[i] = cc::mk64(t0[i], t1[i]);
  cc::mad_eps_co(P, c3, m1);     // V = U + S*EPS mod 2^64, carry c3 (weight 2^64 = EPS)
#pragma unroll
  for (int i = 0; i < N; i++) { t0[i] = (uint32_t)P[i]; t1[i] = (uint32_t)(P[i] >> 32); }
  // Adding or subtracting EPS = 2^32 - 1 under a mask m takes TWO instructions, not three, when the 64-bit result
  // is known not to wrap: x + m*EPS = (lo - m) + (hi + m)*2^32, where lo - m borrows (b) only for lo = 0 and then
  // hands the 2^32 back, i.e. hi + (m & ~b); the mask algebra is one scalar instruction.  Likewise
  // x - m*EPS = (lo + m) + (hi - (m & ~carry))*2^32.
  cc::subb0_co(t0, b3, c3);      // T = V + c3*EPS (cannot wrap: then V < 2^64 - 2^33)
#pragma unroll
  for (int i = 0; i < N; i++) c3[i] &= ~b3[i];
  cc::addc0_cv(t1, c3);
  cc::subb_co(t0, bw, q1, c2);   // u = T - K - c2, borrow bw2
  cc::subb0_co(t1, bw2, bw);
  cc::addc0_co(t0, b3, bw2);     // u - bw2*EPS (cannot borrow: then u >= 2^64 - 2^32)
#pragma unroll
  for (int i = 0; i < N; i++) bw2[i] &= ~b3[i];
  cc::subb0_cv(t1, bw2);
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = cc::mk64(t0[i], t1[i]);
}
#endif
#if defined(__HIP__)
// N (= 3 or 4) independent lazy additions / subtractions / canonicalisations, instruction-interleaved like
// mul_n: the same carry chains as add() / sub() / canon(), no s_nop.  a: any u64, b: canonical.
template <int N>
__device__ __forceinline__ void add_n(const uint64_t (&a)[N], const uint64_t (&b)[N], uint64_t (&r)[N]) {
  uint32_t al[N], ah[N], bl[N], bh[N], lo[N], hi[N];
  cc::mask c1[N], c2[N], c3[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    al[i] = (uint32_t)a[i]; ah[i] = (uint32_t)(a[i] >> 32); bl[i] = (uint32_t)b[i]; bh[i] = (uint32_t)(b[i] >> 32);
  }
  cc::add_co_o(lo, c1, al, bl);      // out of place: a and b stay live for the caller without copies
  cc::addc_co_o(hi, c2, ah, bh, c1);
  cc::subb0_co(lo, c3, c2);          // + c2*EPS in two instructions (mul_n): the sum cannot wrap again
#pragma unroll
  for (int i = 0; i < N; i++) c2[i] &= ~c3[i];
  cc::addc0_cv(hi, c2);
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = cc::mk64(lo[i], hi[i]);
}
template <int N>
__device__ __forceinline__ void sub_n(const uint64_t (&a)[N], const uint64_t (&b)[N], uint64_t (&r)[N]) {
  uint32_t al[N], ah[N], bl[N], bh[N], lo[N], hi[N];
  cc::mask b1[N], b2[N], b3[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    al[i] = (uint32_t)a[i]; ah[i] = (uint32_t)(a[i] >> 32); bl[i] = (uint32_t)b[i]; bh[i] = (uint32_t)(b[i] >> 32);
  }
  cc::sub_co_o(lo, b1, al, bl);
  cc::subb_co_o(hi, b2, ah, bh, b1);
  cc::addc0_co(lo, b3, b2);          // - b2*EPS in two instructions (mul_n): the difference cannot borrow again
#pragma unroll
  for (int i = 0; i < N; i++) b2[i] &= ~b3[i];
  cc::subb0_cv(hi, b2);
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = cc::mk64(lo[i], hi[i]);
}
template <int N>
__device__ __forceinline__ void canon_n(uint64_t (&a)[N]) {
  uint32_t lo[N], hi[N], tl[N], th[N];
  cc::mask k[N], c[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    lo[i] = (uint32_t)a[i]; hi[i] = (uint32_t)(a[i] >> 32);
  }
  cc::add_m1_co_o(tl, k, lo);
  cc::addc0_co_o(th, c, hi, k);
  cc::sel(lo, tl, c);
  cc::sel(hi, th, c);
#pragma unroll
  for (int i = 0; i < N; i++) a[i] = cc::mk64(lo[i], hi[i]);
}
#endif
#if defined(__HIP__)
// Unreduced dot-product accumulator: sum of 64x64-bit products kept as three 64-bit columns
// (a0*b0 | a0*b1 + a1*b0 | a1*b1) plus a wrap counter per column; 4 multiply-adds and 4 carry counts
// per term, one reduction at the end (instead of multiply + reduce + canonical add = 28 per term).
struct DotAcc {
  uint64_t p, m, q;        // column sums modulo 2^64
  uint32_t cp, cm, cq;     // how often each wrapped
};
__device__ __forceinline__ DotAcc dot_zero() { return DotAcc{0, 0, 0, 0, 0, 0}; }
// acc[i] += a[i] * b[i] for four independent accumulators (instruction-interleaved, no s_nop)
__device__ __forceinline__ void dot_mad4(DotAcc (&acc)[4], const uint64_t (&a)[4], const uint64_t (&b)[4]) {
  uint32_t a0[4], a1[4], b0[4], b1[4], cnt[4];
  uint64_t col[4];
  cc::mask c[4], cx[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    a0[i] = (uint32_t)a[i]; a1[i] = (uint32_t)(a[i] >> 32); b0[i] = (uint32_t)b[i]; b1[i] = (uint32_t)(b[i] >> 32);
  }
#define BPG_DOT_COL(COL, CNT, X, Y)                                  \
  _Pragma("unroll") for (int i = 0; i < 4; i++) { col[i] = acc[i].COL; cnt[i] = acc[i].CNT; } \
  cc::mad_co(col, c, X, Y);                                           \
  cc::addc0_co(cnt, cx, c);                                           \
  _Pragma("unroll") for (int i = 0; i < 4; i++) { acc[i].COL = col[i]; acc[i].CNT = cnt[i]; }
  BPG_DOT_COL(p, cp, a0, b0)
  BPG_DOT_COL(m, cm, a0, b1)
  BPG_DOT_COL(m, cm, a1, b0)
  BPG_DOT_COL(q, cq, a1, b1)
#undef BPG_DOT_COL
}
GL_HD uint64_t mulc(uint64_t a, uint64_t b);
GL_HD uint64_t addc(uint64_t a, uint64_t b);
// value = p + (m << 32) + (q << 64) + cp*2^64 + cm*2^96 + cq*2^128  (mod p), canonical
__device__ __forceinline__ uint64_t dot_reduce(const DotAcc& d) {
  uint64_t r = canon(d.p);
  r = addc(r, mulc(d.m, (uint64_t)1 << 32));
  r = addc(r, mulc(d.q, EPS));                       // 2^64 = EPS
  r = addc(r, mulc(d.cp, EPS));
  r = subc(r, canon((uint64_t)d.cm));                // 2^96 = -1
  r = subc(r, mulc(d.cq, (uint64_t)1 << 32));        // 2^128 = -2^32
  return r;
}
#endif
GL_HD uint64_t sqr(uint64_t a) { return mul(a, a); }
GL_HD uint64_t mulc(uint64_t a, uint64_t b) { return canon(mul(a, b)); }

GL_HD uint64_t pow(uint64_t a, uint64_t e) {
  uint64_t r = 1;
  while (e) {
    if (e & 1) r = mulc(r, a);
    a = mulc(a, a);
    e >>= 1;
  }
  return r;
}
// a^(p-2) with p - 2 = (2^32 - 2) * 2^32 + (2^32 - 1): 63 squarings and 9 multiplies instead of the
// 64 + 62 of the generic ladder (same canonical value; inv(0) = 0 as before).
GL_HD uint64_t inv(uint64_t a) {
  // e_k = a^(2^k - 1): e_{j+k} = e_j^(2^k) * e_k
  auto sq_n = [](uint64_t x, int n) {
    for (int i = 0; i < n; i++) x = mul(x, x);
    return x;
  };
  const uint64_t e1 = a;
  const uint64_t e2 = mul(sq_n(e1, 1), e1);
  const uint64_t e3 = mul(sq_n(e2, 1), e1);
  const uint64_t e6 = mul(sq_n(e3, 3), e3);
  const uint64_t e12 = mul(sq_n(e6, 6), e6);
  const uint64_t e24 = mul(sq_n(e12, 12), e12);
  const uint64_t e30 = mul(sq_n(e24, 6), e6);
  const uint64_t e31 = mul(sq_n(e30, 1), e1);
  const uint64_t hi = sq_n(e31, 1);   // a^(2^32 - 2)
  const uint64_t lo = mul(hi, a);     // a^(2^32 - 1)
  return mulc(sq_n(hi, 32), lo);
}
// primitive 2^k-th root of unity (plonky2_field convention)
GL_HD uint64_t root(unsigned k) {
  uint64_t r = TWO_ADIC_ROOT;
  for (unsigned i = k; i < 32; i++) r = mulc(r, r);
  return r;
}

// ---- quadratic extension, components canonical ----
struct Ext {
  uint64_t c0, c1;
};
GL_HD Ext ext(uint64_t a, uint64_t b = 0) { return Ext{a, b}; }
GL_HD bool eq(Ext a, Ext b) { return a.c0 == b.c0 && a.c1 == b.c1; }
GL_HD Ext add(Ext a, Ext b) { return Ext{addc(a.c0, b.c0), addc(a.c1, b.c1)}; }
GL_HD Ext sub(Ext a, Ext b) { return Ext{subc(a.c0, b.c0), subc(a.c1, b.c1)}; }
// 7 * x for any u64 x -> reduced: x*7 = (P0, M0) + M1*2^64 with M1 < 8
GL_HD uint64_t mul7(uint64_t x) {
  const uint64_t p = (uint64_t)(uint32_t)x * 7u;
  const uint64_t m = (x >> 32) * 7u + (p >> 32);
  const uint64_t u = (m << 32) | (uint32_t)p;
  const uint64_t t = u + (m >> 32) * EPS;
  return t < u ? t + EPS : t;
}
GL_HD Ext mul(Ext a, Ext b) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint64_t x[4] = {a.c0, a.c1, a.c0, a.c1}, y[4] = {b.c0, b.c1, b.c1, b.c0};
  uint64_t r[4];
  mul_n<4>(x, y, r);
  return Ext{addc(canon(r[0]), canon(mul7(r[1]))), addc(canon(r[2]), canon(r[3]))};
#else
  uint64_t c0 = addc(mulc(a.c0, b.c0), mulc(W, mulc(a.c1, b.c1)));
  uint64_t c1 = addc(mulc(a.c0, b.c1), mulc(a.c1, b.c0));
  return Ext{c0, c1};
#endif
}
GL_HD Ext scale(Ext a, uint64_t s) { return Ext{mulc(a.c0, s), mulc(a.c1, s)}; }
GL_HD Ext pow(Ext a, uint64_t e) {
  Ext r = ext(1);
  while (e) {
    if (e & 1) r = mul(r, a);
    a = mul(a, a);
    e >>= 1;
  }
  return r;
}
GL_HD Ext inv(Ext a) {
  uint64_t n = subc(mulc(a.c0, a.c0), mulc(W, mulc(a.c1, a.c1)));
  uint64_t ni = inv(n);
  return Ext{mulc(a.c0, ni), mulc(negc(a.c1), ni)};
}

GL_HD uint32_t bitrev(uint32_t x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
  uint32_t r = 0;
  for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
#endif
}

}  // namespace gl
