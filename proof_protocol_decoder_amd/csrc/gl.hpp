// gl.hpp -- Goldilocks field (p = 2^64 - 2^32 + 1) and its quadratic extension for gfx950 device
// code and the C++ host side of the prover (challenger, verifier).
//
// Field / hash / extension degree are the ones fixed by the reference's type aliases:
// plonky_block_proof_gen/src/types.rs:10-18 (GoldilocksField, PoseidonGoldilocksConfig, D = 2).
// Representation: every value stored to memory is canonical (< p).  In registers a value is any
// u64 congruent to the element ("reduced"), and each helper states what it needs and returns.
//
// CDNA4 notes: there is no 64x64 multiplier; a 64x64->128 product is four v_mad_u64_u32 (quarter
// rate) and the Goldilocks reduction (2^64 = 2^32 - 1, 2^96 = -1) is ~10 full-rate VALU ops, so a
// modular multiply is ~25-30 VALU issue slots.  No MFMA anywhere (integer field work).
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

GL_HD uint64_t canon(uint64_t a) { return a >= P ? a - P : a; }

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
GL_HD uint64_t subc(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (P - b); }
GL_HD uint64_t negc(uint64_t a) { return a ? P - a : 0; }

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
  uint64_t lo, hi;
  mul_wide(a, b, lo, hi);
  return reduce128(lo, hi);
}
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
GL_HD uint64_t inv(uint64_t a) { return pow(a, P - 2); }
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
GL_HD Ext mul(Ext a, Ext b) {
  uint64_t c0 = addc(mulc(a.c0, b.c0), mulc(W, mulc(a.c1, b.c1)));
  uint64_t c1 = addc(mulc(a.c0, b.c1), mulc(a.c1, b.c0));
  return Ext{c0, c1};
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
