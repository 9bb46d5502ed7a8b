// air.hpp -- the AIRs this library can prove, as code shared by the device quotient kernel (K5) and the host verifier.
//
// What upstream calls a `Stark` (plonky2_evm: eval_packed_generic / eval_ext, reached from
// plonky_block_proof_gen/src/proof_gen.rs:44-52 through prove_single_table / compute_quotient_polys) is here an
// `air_id` plus ONE templated evaluation routine: the same source is instantiated over canonical base-field words on
// the device (every point of the LDE coset) and over the quadratic extension on the host (the verifier's check at
// zeta).  An AIR is described by
//   * its shape (columns, preprocessed constant columns, constraint degree),
//   * its constraint list: every constraint has a fixed INDEX i in [0, n_constraints), a kind
//     (all rows / transition = times (x - g^(n-1)) / first row = times L_0 / last row = times L_(n-1)) and a degree,
//   * its units: disjoint slices of the list that can be evaluated independently (the kernel spreads the units of a
//     short table over workgroups).
// The random linear combination of starky's ConstraintConsumer, acc = acc * alpha + c in list order, equals
// sum_i c_i * alpha^(T-1-i) with T the total number of constraints (AIR + cross-table-lookup part); the evaluators
// hand (index, value) pairs to the consumer, which weighs them with a table of alpha powers, so a unit may emit its
// constraints in whatever order keeps its operands in registers and the partial sums of units simply add.
//
// AIR 0  synthetic     DESIGN.md section 4 (any width; groups of four columns; degree 3 * deg_pow)
// AIR 1  keccak_f      one round of Keccak-f[1600] per row, 24 rows per permutation, 2431 columns, degree 3;
//                      written from the public specification (FIPS 202 / the Keccak reference), in the style of
//                      upstream's keccak table (~2.4 k columns, constants.rs:12) but NOT upstream's column layout,
//                      which nothing under /root/reference shows [UPSTREAM-UNVERIFIED].
// AIR 2  logic         bitwise AND / OR / XOR of two 256-bit words per row (the zkEVM's logic table, prover_state.rs:85-93
//                      "logic"), 523 columns, degree 3; written from the definition of the three operations, NOT
//                      upstream's column layout [UPSTREAM-UNVERIFIED].
// AIR 3  memory        a memory log sorted by (address, timestamp), one operation per row, 45 columns (44 + the lookup filter), degree 3: a read
//                      returns what the previous operation on the address left there (zero for a first access);
//                      the ordering is enforced by a 32-bit decomposition of the gap to the next row.  In the style
//                      of the zkEVM's memory table (prover_state.rs:85-93 "memory"), its own layout
//                      [UPSTREAM-UNVERIFIED].
// AIR 4  arithmetic    ADD / SUB / LT / GT on 256-bit words as sixteen 16-bit limbs with a carry chain, 309 columns,
//                      degree 2; the additive part of the zkEVM's arithmetic table (prover_state.rs:85-93
//                      "arithmetic"), its own layout [UPSTREAM-UNVERIFIED].
// AIR 5  byte_packing  a big-endian sequence of up to 32 bytes <-> one 256-bit word per row (the zkEVM's byte-packing
//                      table, prover_state.rs:85-93 "byte_packing"), 299 columns, degree 2; its own layout
//                      [UPSTREAM-UNVERIFIED].
// AIR 6  keccak_sponge the absorbing side of Keccak-256: one 136-byte block per row, XORed into the rate, chained from
//                      row to row, pad10*1 on a message's last block (the zkEVM's Keccak sponge table,
//                      prover_state.rs:85-93 "keccak_sponge"), 2414 columns, degree 2; its own layout
//                      [UPSTREAM-UNVERIFIED].  The permutation itself is the Keccak-f table's (AIR 1): this table
//                      carries its input (xored rate, capacity) and its output as columns for a cross-table lookup.
// AIR 7  arithmetic_mul  the multiplicative half of the arithmetic table: x * y = z + 2^256 w on 256-bit words, a 32-column
//                      schoolbook product over sixteen 16-bit limbs with 21-bit carries, 1217 columns, degree 3.  A
//                      table of its own here: merged into AIR 4's rows it would add 900 columns to every addition.
// The auxiliary columns of a table are its cross-table lookups (namespace ctl below; upstream's CTL checks sit beside
// Stark::eval the same way): their constraints follow the AIR's in the list.
#pragma once
#include <cstdint>
#include "gl.hpp"
#if defined(__HIP__)
#include "poseidon.cuh"  // the device form of the Poseidon gate's round function (PoseidonOps<uint64_t>)
#endif

namespace bpg {
namespace air {

constexpr uint32_t SYNTHETIC = 0, KECCAK_F = 1, LOGIC = 2, MEMORY = 3, ARITHMETIC = 4, BYTE_PACKING = 5, KECCAK_SPONGE = 6, ARITHMETIC_MUL = 7, PLONK = 8, COUNT = 9;

struct Shape {
  uint32_t air_id, n_cols, n_const, deg_pow;
};

// ------------------------------------------------------------------------------------------ field policies
// Ops<T>: the arithmetic an evaluator needs.  T = uint64_t: canonical base-field words (device).  T = gl::Ext: host.
template <class T>
struct Ops;

#if defined(__HIP__)
template <>
struct Ops<uint64_t> {
  typedef uint64_t T;
  static __device__ __forceinline__ T k(uint64_t c) { return c; }  // a constant < p
  static __device__ __forceinline__ T add(T a, T b) { return gl::addc(a, b); }
  static __device__ __forceinline__ T sub(T a, T b) { return gl::subc(a, b); }
  static __device__ __forceinline__ T mul(T a, T b) { return gl::mulc(a, b); }
  static __device__ __forceinline__ T dbl(T a) { return gl::addc(a, a); }
  // four independent products, instruction-interleaved carry chains (gl::mul_n), canonical results
  static __device__ __forceinline__ void mul4(const T (&a)[4], const T (&b)[4], T (&r)[4]) {
    // fenced: the scheduler would otherwise overlap two groups and double the live carry masks (SGPR pairs);
    // a mask spilled to a VGPR lane right after the asm that wrote it is a hazard the compiler cannot see
    __builtin_amdgcn_sched_barrier(0);
    gl::mul_n<4>(a, b, r);
    gl::canon_n<4>(r);
    __builtin_amdgcn_sched_barrier(0);
  }
  // The same without canonical results, for products that only feed further products: operands and results are ANY
  // u64 representative (gl::mul_n takes them); add_lazy(a, b) adds a canonical b to such a value, mul7 multiplies one
  // by 7.  A value made this way must pass through mul4 / mul (canonical results) before add / sub / out see it.
  static __device__ __forceinline__ void mul4_lazy(const T (&a)[4], const T (&b)[4], T (&r)[4]) {
    __builtin_amdgcn_sched_barrier(0);
    gl::mul_n<4>(a, b, r);
    __builtin_amdgcn_sched_barrier(0);
  }
  static __device__ __forceinline__ T add_lazy(T a_any, T b) { return gl::add(a_any, b); }
  static __device__ __forceinline__ T sub_lazy(T a_any, T b) { return gl::sub(a_any, b); }
  static __device__ __forceinline__ T mul7(T a_any) { return gl::mul7(a_any); }
};
#endif
template <>
struct Ops<gl::Ext> {
  typedef gl::Ext T;
  static T k(uint64_t c) { return gl::ext(c); }
  static T add(T a, T b) { return gl::add(a, b); }
  static T sub(T a, T b) { return gl::sub(a, b); }
  static T mul(T a, T b) { return gl::mul(a, b); }
  static T dbl(T a) { return gl::add(a, a); }
  static void mul4(const T (&a)[4], const T (&b)[4], T (&r)[4]) {
    for (int i = 0; i < 4; i++) r[i] = gl::mul(a[i], b[i]);
  }
  static void mul4_lazy(const T (&a)[4], const T (&b)[4], T (&r)[4]) { mul4(a, b, r); }
  static T add_lazy(T a, T b) { return gl::add(a, b); }
  static T sub_lazy(T a, T b) { return gl::sub(a, b); }
  static T mul7(T a) { return gl::scale(a, 7); }
};

// ------------------------------------------------------------------------------------------ Poseidon, as constraints
// The round function of Poseidon-Goldilocks (width 12, x^7, MDS = circulant(17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20)
// + diag(8, 0, ...), constants poseidon_rc.inc: the permutation the Merkle trees and the transcript use) over a field
// policy T, for AIR 8's Poseidon gate: sbox_all = x^7 on the twelve words, mds_rc = the MDS layer followed by the
// constants of round `next` (none for next < 0).  Generic form (the host verifier's extension field); the device's
// base-field form is the hash kernels' own code (poseidon.cuh: grouped carry-chain S-boxes, MDS over 32-bit halves).
GL_HD uint64_t poseidon_rc(uint32_t i) {
  constexpr uint64_t RC[360] = {
#include "poseidon_rc.inc"
  };
  return RC[i];
}
GL_HD uint32_t poseidon_mds_entry(uint32_t r, uint32_t c) {  // out[r] = sum_c M[r][c] in[c]
  constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  return C[(c + 12 - r) % 12] + ((r | c) == 0 ? 8u : 0u);
}
template <class T>
struct PoseidonOps {
  typedef Ops<T> F;
  static GL_HD T sbox(T x) {
    const T x2 = F::mul(x, x), x4 = F::mul(x2, x2), x3 = F::mul(x2, x);
    return F::mul(x3, x4);
  }
  static GL_HD void sbox_all(T (&s)[12]) {
    for (uint32_t i = 0; i < 12; i++) s[i] = sbox(s[i]);
  }
  static GL_HD void mds_rc(T (&s)[12], int next) {
    T o[12];
    for (uint32_t r = 0; r < 12; r++) {
      T acc = F::k(next >= 0 ? poseidon_rc(12 * (uint32_t)next + r) : 0);
      for (uint32_t c = 0; c < 12; c++) acc = F::add(acc, F::mul(F::k(poseidon_mds_entry(r, c)), s[c]));
      o[r] = acc;
    }
    for (uint32_t r = 0; r < 12; r++) s[r] = o[r];
  }
  static GL_HD void mds_rc_word0(T (&s)[12], int next) { mds_rc(s, next); }  // (the device form: only s[0] canonical after it)
  static GL_HD T add_rc0(T x, uint32_t i) { return F::add(x, F::k(poseidon_rc(i))); }
};
#if defined(__HIP__)
template <>
struct PoseidonOps<uint64_t> {  // canonical words in, canonical words out (Ops<uint64_t>'s convention)
  static __device__ __forceinline__ uint64_t sbox(uint64_t x) { return gl::canon(poseidon::sbox(x)); }
  static __device__ __forceinline__ void sbox_all(uint64_t (&s)[12]) { poseidon::sbox_all(s); }  // reduced: mds_rc takes any u64
  static __device__ __forceinline__ void mds_rc(uint64_t (&s)[12], int next) {
    if (next >= 0) poseidon::mds<true>(s, poseidon::RC + 12 * next);
    else poseidon::mds<false>(s, nullptr);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]);
  }
  // partial rounds: only word 0 meets a wire before the next layer, and the layer takes any u64
  static __device__ __forceinline__ void mds_rc_word0(uint64_t (&s)[12], int next) {
    poseidon::mds<true>(s, poseidon::RC + 12 * next);
    s[0] = gl::canon(s[0]);
  }
  static __device__ __forceinline__ uint64_t add_rc0(uint64_t x, uint32_t i) { return gl::addc(x, poseidon::RC[i]); }
};
#endif

// ------------------------------------------------------------------------------------------ AIR 0: synthetic
// Per group g of four columns (a, b, c, d), q = constant column g mod K (or 1):
//   index 3g     all rows    c - a*b - q*a
//   index 3g+1   transition  d' - (a*b*c)^e - b
//   index 3g+2   first row   d - a - b
// One unit = `per_unit` consecutive groups.
namespace synthetic {
GL_HD uint32_t n_constraints(const Shape& s) { return 3 * (s.n_cols / 4); }
GL_HD uint32_t per_unit(const Shape& s) {
  const uint32_t G = s.n_cols / 4, u = (G + 47) / 48;
  return u < 8 ? 8 : u;
}
GL_HD uint32_t n_units(const Shape& s) {
  const uint32_t G = s.n_cols / 4, pu = per_unit(s);
  return (G + pu - 1) / pu;
}
template <class T, class Row, class Emit>
GL_HD void eval_unit(const Shape& s, uint32_t unit, const Row& row, Emit& out) {
  typedef Ops<T> F;
  const uint32_t G = s.n_cols / 4, pu = per_unit(s), g0 = unit * pu, g1 = g0 + pu < G ? g0 + pu : G;
#pragma unroll 1
  for (uint32_t g = g0; g < g1; g++) {
    const T a = row.loc(4 * g), b = row.loc(4 * g + 1), c = row.loc(4 * g + 2), d = row.loc(4 * g + 3);
    const T dn = row.nxt(4 * g + 3);
    const T q = s.n_const ? row.cst(g % s.n_const) : F::k(1);
    const T x4[4] = {a, q, a, a}, y4[4] = {b, a, b, b};  // a*b and q*a (two slots spare)
    T p4[4];
    F::mul4(x4, y4, p4);
    const T ab = p4[0], qa = p4[1];
    T t = F::mul(ab, c);
    if (s.deg_pow == 3) t = F::mul(F::mul(t, t), t);
    out.all(3 * g, F::sub(F::sub(c, ab), qa));
    out.transition(3 * g + 1, F::sub(F::sub(dn, t), b));
    out.first(3 * g + 2, F::sub(F::sub(d, a), b));
  }
}
}  // namespace synthetic

// ------------------------------------------------------------------------------------------ AIR 1: Keccak-f[1600]
// Row r of the trace is round r mod 24 of permutation r / 24 (a trace of 2^k rows ends inside a permutation: the
// transition constraints do not apply to the last row).  State lanes A[x][y] (x = column, y = row of the 5 x 5
// sheet, lane index l = x + 5y), bit z of a lane = 2^z.  One round (FIPS 202, section 3.2):
//   theta  C[x] = xor_y A[x][y];  D[x] = C[x-1] ^ rot(C[x+1], 1);  A'[x][y] = A[x][y] ^ D[x]
//   rho/pi B[y][2x+3y] = rot(A'[x][y], R[x][y])
//   chi    A''[x][y] = B[x][y] ^ (~B[x+1][y] & B[x+2][y])
//   iota   A'''[0][0] = A''[0][0] ^ RC[round]
// Columns (all values are field elements; "bits" are 0 / 1 on the trace):
//   0    .. 23     s_i        round flags (one-hot)
//   24   .. 73     A          input lanes as two 32-bit limbs: 24 + 2 l + h
//   74   .. 393    C[x][z]    bits: 74 + 64 x + z
//   394  .. 713    C'[x][z]   bits of C[x] ^ D[x]: 394 + 64 x + z
//   714  .. 2313   A'[l][z]   bits: 714 + 64 l + z
//   2314 .. 2363   A''        limbs: 2314 + 2 l + h
//   2364 .. 2427   A''[0][0]  bits
//   2428 .. 2429   A'''[0][0] limbs
//   2430           g          1 on the last-round row of a permutation the table exposes to the lookup keccak_sponge ->
//                             keccak_f (namespace ctl, which constrains it); committed with the trace, i.e. BEFORE the lookup
//                             challenges are drawn
// Constraints (index ranges; degree; kind):
//   F0  0    .. 23    first row   s_0 - 1, s_i                                                   deg 1
//   F1  24   .. 47    transition  s'_((i+1) mod 24) - s_i                                        deg 1
//   F2  48   .. 2031  all rows    b (b - 1) for the bits of C (320), A' (1600), A''[0][0] (64)   deg 2
//   F3  2032 .. 2351  all rows    C'[x][z] - xor3(C[x][z], C[x-1][z], C[x+1][z-1])               deg 3
//   F4  2352 .. 2671  all rows    d (d - 2)(d - 4), d = sum_y A'[x][y][z] - C'[x][z]             deg 3
//                                 (the column parity of A' is C': together with F5 this forces C = xor_y A)
//   F5  2672 .. 2721  all rows    A limb - sum_z 2^z xor(A'[l][z], C[x][z] ^ C'[x][z])           deg 3
//                                 (A = A' ^ D with D = C ^ C')
//   F6  2722 .. 2771  all rows    A'' limb - sum_z 2^z (B0 ^ (~B1 & B2)),  B from A' by rho / pi deg 3
//   F7  2772 .. 2773  all rows    A''[0][0] limb - sum_z 2^z bit_z                               deg 1
//   F8  2774 .. 2775  all rows    A''' limb - sum_z 2^z xor(bit_z, sum_i s_i RC_i[z])            deg 2
//   F9  2776 .. 2825  transition  (1 - s_23) (A'_next limb - output limb), output = A''' for lane 0 else A''  deg 2
// Units: 0 = F0, F1, F7, F8, F9 and the A''[0][0] bits of F2; 1 + x = the column-x slices of F2 .. F5 (theta); 6 + y = the
// plane-y slice of F6 (chi).
namespace keccak {
constexpr uint32_t N_COLS = 2431, N_CONSTRAINTS = 2826, N_UNITS = 11;
constexpr uint32_t COL_STEP = 0, COL_A = 24, COL_C = 74, COL_CP = 394, COL_AP = 714, COL_APP = 2314, COL_APP0_BITS = 2364,
                   COL_APPP = 2428, COL_G = 2430;  // COL_G: the lookup's filter (namespace ctl), a TRACE column
constexpr uint32_t F0 = 0, F1 = 24, F2 = 48, F3 = 2032, F4 = 2352, F5 = 2672, F6 = 2722, F7 = 2772, F8 = 2774, F9 = 2776;
GL_HD uint64_t round_constant(uint32_t i) {
  constexpr uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
      0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
      0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  return RC[i];
}
// rho offsets R[x][y] (FIPS 202 table 2), indexed by lane x + 5y
GL_HD uint32_t rho(uint32_t l) {
  constexpr uint8_t R[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  return R[l];
}
// B[X][Y] = rot(A'[x][y], R[x][y]) with (X, Y) = (y, 2x + 3y): the source lane of B[X][Y] is x = (X + 3Y) mod 5, y = X
GL_HD uint32_t pi_source(uint32_t X, uint32_t Y) { return (X + 3 * Y) % 5 + 5 * X; }

// a ^ b for 0/1 values as a polynomial: a + b - 2ab, given the product
template <class T>
GL_HD T xor_from_product(T a, T b, T ab) {
  typedef Ops<T> F;
  return F::sub(F::add(a, b), F::dbl(ab));
}

// Loop shape matters on the device: what must stay in registers and is indexed by a loop counter (the limb
// accumulators, the groups of four) sits in fully unrolled loops; the long loops are kept rolled (`unroll 1`), so
// the kernel's register footprint is that of one group of four bits and only a few of the consumer's wave-uniform
// alpha-power loads are in flight at a time (each holds an SGPR pair).
template <class T, class Row, class Emit>
GL_HD void eval_control_unit(const Row& row, Emit& out) {
  typedef Ops<T> F;
  // F0, F1: the round flags start at (1, 0, ..., 0) and rotate
#pragma unroll 1
  for (uint32_t i = 0; i < 24; i++) {
    const T si = row.loc(COL_STEP + i);
    out.first(F0 + i, i == 0 ? F::sub(si, F::k(1)) : si);
    out.transition(F1 + i, F::sub(row.nxt(COL_STEP + (i + 1) % 24), si));
  }
  // A''[0][0] bits: booleanity (F2 tail), limb decomposition (F7), iota (F8)
#pragma unroll 1
  for (uint32_t h = 0; h < 2; h++) {
    T dec = F::k(0), iota = F::k(0);
#pragma unroll 1
    for (uint32_t z0 = 32; z0 > 0; z0 -= 4) {
      T b[4], bb[4];
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) b[i] = row.loc(COL_APP0_BITS + 32 * h + z0 - 1 - i);  // high to low
      F::mul4(b, b, bb);
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) {
        const uint32_t z = 32 * h + z0 - 1 - i;
        out.all(F2 + 1920 + z, F::sub(bb[i], b[i]));
        dec = F::add(F::dbl(dec), b[i]);
        T x = b[i];
        if (((z + 1) & z) == 0) {  // only z = 2^j - 1 is ever set in a round constant
          T rc = F::k(0);          // rc_z = sum of the flags of the rounds whose constant has bit z set
#pragma unroll 1
          for (uint32_t r = 0; r < 24; r++)
            if ((round_constant(r) >> z) & 1) rc = F::add(rc, row.loc(COL_STEP + r));
          x = xor_from_product<T>(b[i], rc, F::mul(b[i], rc));
        }
        iota = F::add(F::dbl(iota), x);
      }
    }
    out.all(F7 + h, F::sub(row.loc(COL_APP + h), dec));
    out.all(F8 + h, F::sub(row.loc(COL_APPP + h), iota));
  }
  // F9: the next row's input is this row's output, except across permutations
  const T not_last = F::sub(F::k(1), row.loc(COL_STEP + 23));
#pragma unroll 1
  for (uint32_t l = 0; l < 25; l++)
#pragma unroll 1
    for (uint32_t h = 0; h < 2; h++) {
      const T o = l == 0 ? row.loc(COL_APPP + h) : row.loc(COL_APP + 2 * l + h);
      out.transition(F9 + 2 * l + h, F::mul(not_last, F::sub(row.nxt(COL_A + 2 * l + h), o)));
    }
}

// Units 1..5, one per sheet column x: theta.  F2 (bits of C[x] and of the five A' lanes of the column), F3 (C' from C),
// F4 (column parity of A'), F5 (the input limbs).  Group of four bits by group; every column of the table the unit
// needs -- C[x], C[x-1], C[x+1], C'[x], the five A' lanes -- is loaded ONCE and used for everything while it is in
// registers.  (The reuse distance of a second pass, ~400 KB per workgroup with ~100 workgroups sharing an XCD's
// 4 MiB L2, is far beyond the cache: round 3's two passes made every load a trip to HBM, 5.2 x the table's bytes,
// profiles/r4_k5_counters.txt.)  F5 is linear in the bits, so a group's share of the limb sum is handed to the
// consumer as it is formed (the fold acc = sum_i c_i alpha^(T-1-i) adds the shares of one index up): no limb
// accumulator lives across groups, and the loop over the five lanes stays rolled.
template <class T, class Row, class Emit>
GL_HD void eval_column_unit(uint32_t x, const Row& row, Emit& out) {
  typedef Ops<T> F;
  const uint32_t xm = (x + 4) % 5, xp = (x + 1) % 5;
#pragma unroll 1
  for (uint32_t y = 0; y < 5; y++)
#pragma unroll 1
    for (uint32_t h = 0; h < 2; h++) out.all(F5 + 2 * (x + 5 * y) + h, row.loc(COL_A + 2 * (x + 5 * y) + h));
#pragma unroll 1
  for (uint32_t zt = 63; zt < 64; zt -= 4) {  // the group is bits zt, zt - 1, zt - 2, zt - 3
    const uint32_t h = zt >> 5;
    const T weight = F::k((uint64_t)1 << ((zt - 3) & 31));  // 2^(lowest bit of the group within its limb)
    T c[4], cp[4], dd[4], u[4];
    {
      T cm[4], cq[4], t1[4], t2[4];
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) {
        c[i] = row.loc(COL_C + 64 * x + zt - i);
        cp[i] = row.loc(COL_CP + 64 * x + zt - i);
        cm[i] = row.loc(COL_C + 64 * xm + zt - i);
        cq[i] = row.loc(COL_C + 64 * xp + (zt - i + 63) % 64);
      }
      // C' = xor3(C, C[x-1], rot(C[x+1], 1)) = u + w - 2uw with u = C ^ C[x-1], w = the rotated bit
      F::mul4(c, c, t1);
      F::mul4(c, cm, t2);
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) {
        out.all(F2 + 64 * x + zt - i, F::sub(t1[i], c[i]));
        u[i] = xor_from_product<T>(c[i], cm[i], t2[i]);
      }
      F::mul4(u, cq, t1);
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) out.all(F3 + 64 * x + zt - i, F::sub(cp[i], xor_from_product<T>(u[i], cq[i], t1[i])));
      F::mul4(c, cp, t2);  // D = C ^ C'
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) {
        dd[i] = xor_from_product<T>(c[i], cp[i], t2[i]);
        u[i] = F::sub(F::k(0), cp[i]);  // becomes d = sum_y A'[x][y][z] - C'[x][z]
      }
    }
#pragma unroll 1
    for (uint32_t y = 0; y < 5; y++) {  // the five lanes of the column: booleanity, parity, A = A' ^ D
      T ap[4], apap[4], apd[4];
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) ap[i] = row.loc(COL_AP + 64 * (x + 5 * y) + zt - i);
      F::mul4(ap, ap, apap);
      F::mul4(ap, dd, apd);
      T g = F::k(0);  // the group's four bits of A = A' ^ D, bit zt on top
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) {
        out.all(F2 + 320 + 64 * (x + 5 * y) + zt - i, F::sub(apap[i], ap[i]));
        u[i] = F::add(u[i], ap[i]);
        g = F::add(F::dbl(g), xor_from_product<T>(ap[i], dd[i], apd[i]));
      }
      out.all(F5 + 2 * (x + 5 * y) + h, F::sub(F::k(0), F::mul(g, weight)));
    }
    {
      T t1[4], t2[4], p[4], q[4];
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) {
        t1[i] = F::sub(u[i], F::k(2));
        t2[i] = F::sub(u[i], F::k(4));
      }
      F::mul4(u, t1, p);
      F::mul4(p, t2, q);
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) out.all(F4 + 64 * x + zt - i, q[i]);
    }
  }
}

// Units 6..10, one per sheet plane y: chi (F6) on the five output lanes (x, y).  B[.][y] draws on exactly five A'
// lanes, one from every plane ((x + 3y) mod 5, x), each rotated by its own offset.  A group of four bits of all five
// B lanes is loaded once and serves the five outputs from registers (each B is b0 of one output, b1 of another, b2 of a
// third); the outputs' limb sums go to the consumer group by group, as in the column units.
template <class T, class Row, class Emit>
GL_HD void eval_plane_unit(uint32_t y, const Row& row, Emit& out) {
  typedef Ops<T> F;
  uint32_t src[5], rot[5];
#pragma unroll
  for (uint32_t x = 0; x < 5; x++) {
    src[x] = pi_source(x, y);
    rot[x] = rho(src[x]);
  }
#pragma unroll 1
  for (uint32_t l = 0; l < 5; l++)
#pragma unroll 1
    for (uint32_t h = 0; h < 2; h++) out.all(F6 + 2 * (l + 5 * y) + h, row.loc(COL_APP + 2 * (l + 5 * y) + h));
#pragma unroll 1
  for (uint32_t zt = 63; zt < 64; zt -= 4) {
    const uint32_t h = zt >> 5;
    const T weight = F::k((uint64_t)1 << ((zt - 3) & 31));
    T b[5][4];  // B[x][z] = A'[source of x][(z - rho) mod 64] for z = zt - i
#pragma unroll
    for (uint32_t x = 0; x < 5; x++)
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) b[x][i] = row.loc(COL_AP + 64 * src[x] + (zt - i + 64 - rot[x]) % 64);
#pragma unroll
    for (uint32_t x = 0; x < 5; x++) {
      T nb1[4], t[4], bt[4];
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) nb1[i] = F::sub(F::k(1), b[(x + 1) % 5][i]);
      F::mul4(nb1, b[(x + 2) % 5], t);
      F::mul4(b[x], t, bt);
      T g = F::k(0);
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) g = F::add(F::dbl(g), xor_from_product<T>(b[x][i], t[i], bt[i]));
      out.all(F6 + 2 * (x + 5 * y) + h, F::sub(F::k(0), F::mul(g, weight)));
    }
  }
}

template <class T, class Row, class Emit>
GL_HD void eval_unit(uint32_t unit, const Row& row, Emit& out) {
  if (unit == 0) eval_control_unit<T>(row, out);
  else if (unit <= 5) eval_column_unit<T>(unit - 1, row, out);
  else eval_plane_unit<T>(unit - 6, row, out);
}

// One round on 25 lanes; optionally the intermediate values a trace row needs.
struct Round {
  uint64_t c[5], cp[5], ap[25], app[25], appp0;
};
GL_HD void round(const uint64_t a[25], uint32_t rnd, Round& o) {
  for (uint32_t x = 0; x < 5; x++) o.c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
  for (uint32_t x = 0; x < 5; x++) {
    const uint64_t n = o.c[(x + 1) % 5];
    const uint64_t d = o.c[(x + 4) % 5] ^ ((n << 1) | (n >> 63));
    o.cp[x] = o.c[x] ^ d;
    for (uint32_t y = 0; y < 5; y++) o.ap[x + 5 * y] = a[x + 5 * y] ^ d;
  }
  uint64_t b[25];
  for (uint32_t X = 0; X < 5; X++)
    for (uint32_t Y = 0; Y < 5; Y++) {
      const uint32_t l = pi_source(X, Y), r = rho(l);
      b[X + 5 * Y] = r ? ((o.ap[l] << r) | (o.ap[l] >> (64 - r))) : o.ap[l];
    }
  for (uint32_t y = 0; y < 5; y++)
    for (uint32_t x = 0; x < 5; x++) o.app[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
  o.appp0 = o.app[0] ^ round_constant(rnd);
}
}  // namespace keccak

// ------------------------------------------------------------------------------------------ AIR 2: logic
// One operation per row on two 256-bit words given as bits; the result as eight 32-bit limbs (the form in which the
// other tables of a zkEVM would look it up).  A row whose three flags are all zero is padding: its result is zero.
// Columns:
//   0 .. 2       is_and, is_or, is_xor
//   3 .. 258     input 0 bits: 3 + 32 k + z   (limb k, bit z)
//   259 .. 514   input 1 bits: 259 + 32 k + z
//   515 .. 522   result limbs
//   523          g: 1 on the operations the table exposes to the lookup keccak_sponge -> logic (namespace ctl)
// Constraints (all rows):
//   L0  0 .. 2       f (f - 1) for the three flags                                               deg 2
//   L1  3            s (s - 1), s = is_and + is_or + is_xor: at most one operation               deg 2
//   L2  4 .. 515     b (b - 1) for the input bits: 4 + 256 j + 32 k + z                          deg 2
//   L3  516 .. 523   result_k - sum_z 2^z (p (a_z + b_z) + q a_z b_z),  p = is_or + is_xor,      deg 3
//                    q = is_and - is_or - 2 is_xor   (and: ab, or: a + b - ab, xor: a + b - 2ab)
// Units: unit k = the bits of limb k of both inputs (L2) and L3 + k; unit 0 also takes L0 and L1.
namespace logic {
constexpr uint32_t N_COLS = 524, N_CONSTRAINTS = 524, N_UNITS = 8;
constexpr uint32_t COL_OP = 0, COL_IN0 = 3, COL_IN1 = 259, COL_RES = 515, COL_G = 523;  // COL_G: the lookup's filter (namespace ctl), a TRACE column
constexpr uint32_t L0 = 0, L1 = 3, L2 = 4, L3 = 516;
constexpr uint32_t OP_NONE = 0, OP_AND = 1, OP_OR = 2, OP_XOR = 3;
GL_HD uint32_t apply(uint32_t op, uint32_t a, uint32_t b) {
  return op == OP_AND ? (a & b) : op == OP_OR ? (a | b) : op == OP_XOR ? (a ^ b) : 0u;
}
// The unit of limb k also hands the lookup keccak_sponge -> logic (namespace ctl) the limb's SHARE of the two product
// constraints: a product's term 1 + g (gamma + v - 1) is linear in the tuple v = sum_j beta^j t_j, so the part of
// z - z' term that carries (input-0 limb k, input-1 limb k, result limb k) is emitted here, where the limb sums exist
// anyway, under the constraint's own index (the consumer adds the shares of one index up); the rest -- z - z' (1 + g
// (gamma + flags - 1)) -- is ctl::eval's.  ctl_base: the index of the table's first lookup constraint (the filter bit).
template <class T, class Row, class Emit>
GL_HD void eval_unit(uint32_t k, uint32_t ctl_base, const uint64_t ctl[4], const Row& row, Emit& out) {
  typedef Ops<T> F;
  const T f_and = row.loc(COL_OP), f_or = row.loc(COL_OP + 1), f_xor = row.loc(COL_OP + 2);
  if (k == 0) {
    const T s = F::add(F::add(f_and, f_or), f_xor);
    const T x4[4] = {f_and, f_or, f_xor, s};
    T xx[4];
    F::mul4(x4, x4, xx);
#pragma unroll
    for (uint32_t i = 0; i < 3; i++) out.all(L0 + i, F::sub(xx[i], x4[i]));
    out.all(L1, F::sub(xx[3], s));
  }
  T sum = F::k(0), prod = F::k(0), la = F::k(0);  // sum_z 2^z (a_z + b_z), sum_z 2^z a_z b_z and sum_z 2^z a_z, Horner from the top bit down
#pragma unroll 1
  for (uint32_t z0 = 32; z0 > 0; z0 -= 4) {
    T a[4], b[4], aa[4], bb[4], ab[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      a[i] = row.loc(COL_IN0 + 32 * k + z0 - 1 - i);
      b[i] = row.loc(COL_IN1 + 32 * k + z0 - 1 - i);
    }
    F::mul4(a, a, aa);
    F::mul4(b, b, bb);
    F::mul4(a, b, ab);
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      const uint32_t z = z0 - 1 - i;
      out.all(L2 + 32 * k + z, F::sub(aa[i], a[i]));
      out.all(L2 + 256 + 32 * k + z, F::sub(bb[i], b[i]));
      sum = F::add(F::dbl(sum), F::add(a[i], b[i]));
      prod = F::add(F::dbl(prod), ab[i]);
      la = F::add(F::dbl(la), a[i]);
    }
  }
  const T p = F::add(f_or, f_xor), q = F::sub(F::sub(f_and, f_or), F::dbl(f_xor));
  const T res = row.loc(COL_RES + k);
  out.all(L3 + k, F::sub(res, F::add(F::mul(p, sum), F::mul(q, prod))));
  // the limb's share of the lookup: tuple positions 3 + k (input 0), 11 + k (input 1), 19 + k (result)
  const T g = row.loc(COL_G), lb = F::sub(sum, la);
#pragma unroll 1
  for (uint32_t c = 0; c < 2; c++) {
    const T beta = F::k(ctl[2 * c]), b8 = F::k(gl::pow(ctl[2 * c], 8)), bk = F::k(gl::pow(ctl[2 * c], 3 + k));
    (void)beta;
    const T x = F::mul(bk, F::add(la, F::mul(b8, F::add(lb, F::mul(b8, res)))));
    const T gx = F::mul(g, x);
    out.transition(ctl_base + 1 + 2 * c, F::sub(F::k(0), F::mul(row.aux_nxt(c), gx)));
    out.last(ctl_base + 2 + 2 * c, F::sub(F::k(0), gx));
  }
}
}  // namespace logic

// ------------------------------------------------------------------------------------------ AIR 3: memory
// Rows are the operations of a memory log sorted by (address, timestamp); addresses and timestamps are below 2^32.
// Columns:
//   0          is_read (0 = write)
//   1          address
//   2          timestamp
//   3 .. 10    value, eight 32-bit limbs (their range is the business of the table that looks the value up)
//   11         address_changed: the NEXT row is on another address
//   12 .. 43   bits of the gap to the next row: address' - address - 1 if address_changed, else timestamp' - timestamp - 1
//   44         g: 1 on the operations the table exposes to the lookup byte_packing -> memory (namespace ctl, which
//              constrains it); committed with the trace, i.e. before the lookup challenges are drawn
// Constraints:
//   M0  0          all rows    is_read (is_read - 1)                                             deg 2
//   M1  1          all rows    address_changed (address_changed - 1)                             deg 2
//   M2  2 .. 33    all rows    g (g - 1) for the gap bits                                        deg 2
//   M3  34         transition  (1 - address_changed) (address' - address)                        deg 2
//   M4  35         transition  address_changed (address' - address - 1 - G)
//                              + (1 - address_changed) (timestamp' - timestamp - 1 - G), G = sum_z 2^z g_z   deg 2
//                              (a gap of 32 bits: addresses strictly increase across changes, timestamps within)
//   M5  36 .. 43   transition  (1 - address_changed) is_read' (value'_k - value_k)              deg 3
//   M6  44 .. 51   transition  address_changed is_read' value'_k       (memory starts as zeros)  deg 3
//   M7  52 .. 59   first row   is_read value_k                                                   deg 2
// One unit.
namespace memory {
constexpr uint32_t N_COLS = 45, N_CONSTRAINTS = 60, N_UNITS = 1;
constexpr uint32_t COL_READ = 0, COL_ADDR = 1, COL_TS = 2, COL_VAL = 3, COL_CHG = 11, COL_GAP = 12, COL_G = 44;
constexpr uint32_t M0 = 0, M1 = 1, M2 = 2, M3 = 34, M4 = 35, M5 = 36, M6 = 44, M7 = 52;
// (Sixty constraints on 45 columns: the plain multiply is used outside the bit loop -- the carry-chain groups of four
// would hold three sets of masks next to the column offsets and spill scalars for nothing.)
template <class T, class Row, class Emit>
GL_HD void eval_unit(const Row& row, Emit& out) {
  typedef Ops<T> F;
  const T rd = row.loc(COL_READ), chg = row.loc(COL_CHG), rdn = row.nxt(COL_READ);
  const T same = F::sub(F::k(1), chg);
  out.all(M0, F::sub(F::mul(rd, rd), rd));
  out.all(M1, F::sub(F::mul(chg, chg), chg));
  T gap = F::k(0);
#pragma unroll 1
  for (uint32_t z0 = 32; z0 > 0; z0 -= 4) {
    T g[4], gg[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) g[i] = row.loc(COL_GAP + z0 - 1 - i);
    F::mul4(g, g, gg);
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      out.all(M2 + z0 - 1 - i, F::sub(gg[i], g[i]));
      gap = F::add(F::dbl(gap), g[i]);
    }
  }
  const T da = F::sub(row.nxt(COL_ADDR), row.loc(COL_ADDR)), dt = F::sub(row.nxt(COL_TS), row.loc(COL_TS));
  const T g1 = F::add(gap, F::k(1));
  out.transition(M3, F::mul(same, da));
  out.transition(M4, F::add(F::mul(chg, F::sub(da, g1)), F::mul(same, F::sub(dt, g1))));
  const T same_read = F::mul(same, rdn), new_read = F::mul(chg, rdn);
#pragma unroll 1
  for (uint32_t k = 0; k < 8; k++) {
    const T v = row.loc(COL_VAL + k), vn = row.nxt(COL_VAL + k);
    out.transition(M5 + k, F::mul(same_read, F::sub(vn, v)));
    out.transition(M6 + k, F::mul(new_read, vn));
    out.first(M7 + k, F::mul(rd, v));
  }
}
}  // namespace memory

// ------------------------------------------------------------------------------------------ AIR 4: arithmetic
// One operation per row on 256-bit words x, y given as sixteen 16-bit limbs; a third word z is given by its BITS (so
// its limbs are range-checked here; the range of x and y is the business of the table that supplies them) and every
// operation is one carry chain  U + V = W + 2^256 c_15  over the limbs, with (U, V, W) chosen by the flag:
//   add  x + y = z              (x, y, z)          sub  x - y = z  <=>  z + y = x       (z, y, x)
//   lt   x < y  <=>  x - y borrows: z + y = x + 2^256 result      (z, y, x), result = c_15
//   gt   x > y  <=>  y - x borrows: z + x = y + 2^256 result      (z, x, y), result = c_15
// A row without a flag is padding (its carries are forced to zero).
// Columns:
//   0 .. 3       is_add, is_sub, is_lt, is_gt
//   4 .. 19      x limbs        20 .. 35   y limbs
//   36 .. 291    z bits: 36 + 16 k + j (limb k, bit j)
//   292 .. 307   carry out of limb k
//   308          result (the comparison bit for lt / gt, else 0)
// Constraints (all rows):
//   A0  0 .. 3      f (f - 1)                          A1  4          s (s - 1), s = sum of the flags      deg 2
//   A2  5 .. 260    z bits are bits                    A3  261 .. 276 carries are bits                     deg 2
//   A4  277 .. 292  U_k + V_k + c_(k-1) - W_k - 2^16 c_k   (c_(-1) = 0)                                    deg 2
//   A5  293         result - (is_lt + is_gt) c_15                                                          deg 2
// Units: unit u = limbs 4u .. 4u + 3; unit 0 also takes A0, A1, A5.
namespace arithmetic {
constexpr uint32_t N_COLS = 309, N_CONSTRAINTS = 294, N_UNITS = 4;
constexpr uint32_t COL_OP = 0, COL_X = 4, COL_Y = 20, COL_Z = 36, COL_CARRY = 292, COL_RES = 308;
constexpr uint32_t A0 = 0, A1 = 4, A2 = 5, A3 = 261, A4 = 277, A5 = 293;
constexpr uint32_t OP_NONE = 0, OP_ADD = 1, OP_SUB = 2, OP_LT = 3, OP_GT = 4;
template <class T, class Row, class Emit>
GL_HD void eval_unit(uint32_t u, const Row& row, Emit& out) {
  typedef Ops<T> F;
  const T f_add = row.loc(COL_OP), f_sub = row.loc(COL_OP + 1), f_lt = row.loc(COL_OP + 2), f_gt = row.loc(COL_OP + 3);
  if (u == 0) {
    const T s = F::add(F::add(f_add, f_sub), F::add(f_lt, f_gt));
    const T x4[4] = {f_add, f_sub, f_lt, f_gt};
    T xx[4];
    F::mul4(x4, x4, xx);
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) out.all(A0 + i, F::sub(xx[i], x4[i]));
    out.all(A1, F::sub(F::mul(s, s), s));
    out.all(A5, F::sub(row.loc(COL_RES), F::mul(F::add(f_lt, f_gt), row.loc(COL_CARRY + 15))));
  }
  // the flag sums that select (U, V, W): U = f_add x + fz z,  V = fy y + f_gt x,  W = f_add z + fx x + f_gt y
  const T fz = F::add(F::add(f_sub, f_lt), f_gt), fy = F::add(F::add(f_add, f_sub), f_lt), fx = F::add(f_sub, f_lt);
#pragma unroll 1
  for (uint32_t k = 4 * u; k < 4 * u + 4; k++) {
    T z = F::k(0);
#pragma unroll 1
    for (uint32_t j0 = 16; j0 > 0; j0 -= 4) {
      T b[4], bb[4];
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) b[i] = row.loc(COL_Z + 16 * k + j0 - 1 - i);
      F::mul4(b, b, bb);
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) {
        out.all(A2 + 16 * k + j0 - 1 - i, F::sub(bb[i], b[i]));
        z = F::add(F::dbl(z), b[i]);
      }
    }
    const T x = row.loc(COL_X + k), y = row.loc(COL_Y + k), c = row.loc(COL_CARRY + k);
    // (f_add + f_gt - fx) x + (fy - f_gt) y + (fz - f_add) z  =  U + V - W   (x, y, z each appear on both sides)
    const T a4[4] = {F::sub(F::add(f_add, f_gt), fx), F::sub(fy, f_gt), F::sub(fz, f_add), c}, b4[4] = {x, y, z, c};
    T p4[4];
    F::mul4(a4, b4, p4);
    out.all(A3 + k, F::sub(p4[3], c));
    T e = F::add(F::add(p4[0], p4[1]), p4[2]);
    if (k) e = F::add(e, row.loc(COL_CARRY + k - 1));
    out.all(A4 + k, F::sub(e, F::mul(F::k(65536), c)));
  }
}
}  // namespace arithmetic

// ------------------------------------------------------------------------------------------ AIR 5: byte packing
// One sequence per row: `len` bytes (1..32) read from or written to memory, most significant first, and the 256-bit
// word they are the big-endian form of (what MLOAD_32BYTES / MSTORE_32BYTES move), as eight 32-bit limbs, least
// significant limb first.  A row without a length flag is padding.
// Columns:
//   0            is_read
//   1 .. 32      length flags: column j is 1 when len = j
//   33 .. 288    the bits of the 32 byte slots: 33 + 8 i + b  (slot i = the i-th byte of the sequence)
//   289 .. 296   value limbs
//   297, 298     address, timestamp of the memory operation that moves the word (free columns of the AIR; the lookup
//                byte_packing -> memory, namespace ctl, sends (is_read, address, timestamp, value limbs) to the memory table)
// Constraints (all rows):
//   P0  0          is_read is a bit                         P1  1 .. 32    length flags are bits           deg 2
//   P2  33         at most one length flag                  P3  34 .. 289  slot bits are bits              deg 2
//   P4  290 .. 321 slot i is zero unless i < len:  byte_i (1 - sum_{j > i} flag_j)                         deg 2
//   P5  322 .. 329 value limb k = sum_j flag_j sum_{i < j, 4k <= j-1-i < 4k+4} byte_i 256^(j-1-i-4k)       deg 2
// Units: unit 0 = P0 .. P2; units 1 .. 8 = slots 4(u-1) .. 4(u-1)+3 (P3, P4) and value limb u - 1 (P5).
namespace byte_packing {
constexpr uint32_t N_COLS = 299, N_CONSTRAINTS = 330, N_UNITS = 9;
constexpr uint32_t COL_READ = 0, COL_LEN = 1, COL_BITS = 33, COL_VAL = 289, COL_ADDR = 297, COL_TS = 298;
constexpr uint32_t P0 = 0, P1 = 1, P2 = 33, P3 = 34, P4 = 290, P5 = 322;
// byte slot i from its bits (Horner)
template <class T, class Row>
GL_HD T slot(const Row& row, uint32_t i) {
  typedef Ops<T> F;
  T v = F::k(0);
#pragma unroll 1
  for (uint32_t b = 8; b > 0; b--) v = F::add(F::dbl(v), row.loc(COL_BITS + 8 * i + b - 1));
  return v;
}
template <class T, class Row, class Emit>
GL_HD void eval_unit(uint32_t u, const Row& row, Emit& out) {
  typedef Ops<T> F;
  if (u == 0) {
    const T rd = row.loc(COL_READ);
    out.all(P0, F::sub(F::mul(rd, rd), rd));
    T s = F::k(0);
#pragma unroll 1
    for (uint32_t j = 0; j < 32; j++) {
      const T f = row.loc(COL_LEN + j);
      out.all(P1 + j, F::sub(F::mul(f, f), f));
      s = F::add(s, f);
    }
    out.all(P2, F::sub(F::mul(s, s), s));
    return;
  }
  const uint32_t k = u - 1;
  // slots 4k .. 4k + 3: bits, and "zero unless inside the sequence"
#pragma unroll 1
  for (uint32_t i = 4 * k; i < 4 * k + 4; i++) {
    T v = F::k(0);
#pragma unroll 1
    for (uint32_t b0 = 8; b0 > 0; b0--) {
      const T g = row.loc(COL_BITS + 8 * i + b0 - 1);
      out.all(P3 + 8 * i + b0 - 1, F::sub(F::mul(g, g), g));
      v = F::add(F::dbl(v), g);
    }
    T longer = F::k(0);  // sum of the flags of the lengths that contain slot i: len > i, i.e. flag columns j = i+1 .. 32
#pragma unroll 1
    for (uint32_t j = i + 1; j <= 32; j++) longer = F::add(longer, row.loc(COL_LEN + j - 1));
    out.all(P4 + i, F::mul(v, F::sub(F::k(1), longer)));
  }
  // value limb k = sum_j flag_j sum_{p = 4k .. 4k+3, p < j} slot_(j-1-p) 256^(p-4k), regrouped by slot: slot i meets
  // weight 256^q of limb k under the length j = i + 1 + 4k + q, so
  //   limb k = sum_i slot_i (flag_(i+1+4k) + 256 flag_(i+2+4k) + 256^2 flag_(i+3+4k) + 256^3 flag_(i+4+4k))
  T acc = F::k(0);
#pragma unroll 1
  for (uint32_t i = 0; i + 1 + 4 * k <= 32; i++) {
    T sel = F::k(0);
#pragma unroll
    for (uint32_t q = 4; q > 0; q--) {  // Horner in 256 over the (up to) four lengths, largest weight first
      const uint32_t j = i + 4 * k + q;
      sel = F::mul(sel, F::k(256));
      if (j <= 32) sel = F::add(sel, row.loc(COL_LEN + j - 1));
    }
    acc = F::add(acc, F::mul(slot<T>(row, i), sel));
  }
  out.all(P5 + k, F::sub(row.loc(COL_VAL + k), acc));
}
}  // namespace byte_packing

// ------------------------------------------------------------------------------------------ AIR 6: Keccak sponge
// One absorbed block per row.  A message is a run of rows: zero or more FULL blocks (136 message bytes) and one FINAL
// block (0..135 message bytes, then pad10*1: 0x01 after the last byte, 0x80 on byte 135).  The state before the first
// block of a message is zero; after a full block the next row starts from this row's updated state.  A row with
// neither flag is padding.
// Columns ("limb" = 32 bits; state limb 2 l + h = half h of lane l = x + 5y):
//   0, 1           is_full, is_final
//   2 .. 137       final-length flags: column 2 + j is 1 when the final block holds j message bytes
//   138 .. 1225    bits of the block as absorbed: 138 + 8 i + b (byte i, bit b)
//   1226 .. 2313   bits of the rate part of the state before the block: 1226 + 32 k + z (limb k < 34)
//   2314 .. 2329   capacity limbs of the state before the block
//   2330 .. 2363   rate limbs after the XOR (with the capacity: the input of the permutation)
//   2364 .. 2413   the state after the permutation, 50 limbs -- NOT constrained here: tying (xored rate, capacity) ->
//                  updated state to a permutation is the cross-table lookup into the Keccak-f table
// Constraints:
//   K0  0 .. 1        all rows    flags are bits                K1  2          all rows  at most one flag      deg 2
//   K2  3 .. 138      all rows    length flags are bits         K3  139        all rows  sum of them = is_final deg 2 / 1
//   K4  140 .. 1227   all rows    block bits are bits           K5  1228 .. 2315 all rows rate bits are bits    deg 2
//   K6  2316 .. 2451  all rows    pad10*1: with A_i = sum_{j < i} L_j (byte i lies after the message) and
//                                 c_i = 0x80 for i = 135, else 0:  A_i (byte_i - c_i) + L_i (byte_i - 1 - c_i)  deg 2
//   K7  2452 .. 2485  all rows    xored limb k - sum_z 2^z xor(rate bit, block bit)                             deg 2
//   K8  2486 .. 2535  transition  next row's state before - is_full * updated state (50 limbs)                  deg 2
//   K9  2536 .. 2585  first row   the state before is zero (50 limbs)                                          deg 1
//   K10 2586          transition  is_full (1 - is_full' - is_final'): a message does not end on a full block   deg 2
// Units: 0 = K0 .. K3, K10; 1 + k (k < 34) = rate limb k (bytes 4k .. 4k+3 of the block); 35 = the capacity limbs.
namespace keccak_sponge {
constexpr uint32_t N_COLS = 2414, N_CONSTRAINTS = 2587, N_UNITS = 36;
constexpr uint32_t COL_FULL = 0, COL_FINAL = 1, COL_LEN = 2, COL_BLOCK = 138, COL_RATE = 1226, COL_CAP = 2314, COL_XORED = 2330,
                   COL_UPDATED = 2364;
constexpr uint32_t K0 = 0, K1 = 2, K2 = 3, K3 = 139, K4 = 140, K5 = 1228, K6 = 2316, K7 = 2452, K8 = 2486, K9 = 2536, K10 = 2586;
// (ctl_base, ctl: the limb units also emit their limb's share of the lookup keccak_sponge -> logic, as the logic table's
// limb units do: logic::eval_unit)
template <class T, class Row, class Emit>
GL_HD void eval_unit(uint32_t u, uint32_t ctl_base, const uint64_t ctl[4], const Row& row, Emit& out) {
  typedef Ops<T> F;
  const T full = row.loc(COL_FULL);
  if (u == 0) {
    const T fin = row.loc(COL_FINAL), s = F::add(full, fin);
    out.all(K0, F::sub(F::mul(full, full), full));
    out.all(K0 + 1, F::sub(F::mul(fin, fin), fin));
    out.all(K1, F::sub(F::mul(s, s), s));
    T n = F::k(0);
#pragma unroll 1
    for (uint32_t j = 0; j < 136; j++) {
      const T l = row.loc(COL_LEN + j);
      out.all(K2 + j, F::sub(F::mul(l, l), l));
      n = F::add(n, l);
    }
    out.all(K3, F::sub(n, fin));
    out.transition(K10, F::mul(full, F::sub(F::sub(F::k(1), row.nxt(COL_FULL)), row.nxt(COL_FINAL))));
    return;
  }
  if (u == 35) {
#pragma unroll 1
    for (uint32_t k = 0; k < 16; k++) {
      out.transition(K8 + 34 + k, F::sub(row.nxt(COL_CAP + k), F::mul(full, row.loc(COL_UPDATED + 34 + k))));
      out.first(K9 + 34 + k, row.loc(COL_CAP + k));
    }
    return;
  }
  const uint32_t k = u - 1;
  // (plain multiplies, one bit at a time: the carry-chain groups of four next to five running sums and the rolled
  // loops' column offsets cost this kernel 177 VGPRs and 35 spilled scalars)
  {
    T after = F::k(0);  // A_i: the sum of the length flags below byte i
#pragma unroll 1
    for (uint32_t j = 0; j < 4 * k; j++) after = F::add(after, row.loc(COL_LEN + j));
#pragma unroll 1
    for (uint32_t i = 4 * k; i < 4 * k + 4; i++) {  // the four bytes of the limb: bits, and pad10*1
      T byte = F::k(0);
#pragma unroll 1
      for (uint32_t b = 8; b > 0; b--) {
        const T bb = row.loc(COL_BLOCK + 8 * i + b - 1);
        out.all(K4 + 8 * i + b - 1, F::sub(F::mul(bb, bb), bb));
        byte = F::add(F::dbl(byte), bb);
      }
      const T l_i = row.loc(COL_LEN + i), v = F::sub(byte, F::k(i == 135 ? 0x80 : 0));
      out.all(K6 + i, F::add(F::mul(after, v), F::mul(l_i, F::sub(v, F::k(1)))));
      after = F::add(after, l_i);
    }
  }
  T xored = F::k(0), before = F::k(0), blk = F::k(0);
#pragma unroll 1
  for (uint32_t z = 32; z > 0; z--) {  // limb bit z - 1 = bit (z - 1) % 8 of byte 4k + (z - 1) / 8
    const T bb = row.loc(COL_BLOCK + 32 * k + z - 1), rr = row.loc(COL_RATE + 32 * k + z - 1);
    out.all(K5 + 32 * k + z - 1, F::sub(F::mul(rr, rr), rr));
    xored = F::add(F::dbl(xored), F::sub(F::add(bb, rr), F::dbl(F::mul(bb, rr))));
    before = F::add(F::dbl(before), rr);
    blk = F::add(F::dbl(blk), bb);
  }
  const T xcol = row.loc(COL_XORED + k);
  out.all(K7 + k, F::sub(xcol, xored));
  {
    // the limb's share of the lookup keccak_sponge -> logic: limb group m = k / 8, tuple positions 3 + j (rate before),
    // 11 + j (block), 19 + j (xored), j = k % 8; product column 2 + 2 m + c of the table's lookup constraints
    const uint32_t m = k >> 3, j = k & 7;
    const T f = F::add(full, row.loc(COL_FINAL));
#pragma unroll 1
    for (uint32_t c = 0; c < 2; c++) {
      const T b8 = F::k(gl::pow(ctl[2 * c], 8)), bj = F::k(gl::pow(ctl[2 * c], 3 + j));
      const T x = F::mul(bj, F::add(before, F::mul(b8, F::add(blk, F::mul(b8, xcol)))));
      const T fx = F::mul(f, x);
      const uint32_t col = 2 + 2 * m + c;
      out.transition(ctl_base + 2 * col, F::sub(F::k(0), F::mul(row.aux_nxt(col), fx)));
      out.last(ctl_base + 2 * col + 1, F::sub(F::k(0), fx));
    }
  }
  T before_next = F::k(0);
#pragma unroll 1
  for (uint32_t z = 32; z > 0; z--) before_next = F::add(F::dbl(before_next), row.nxt(COL_RATE + 32 * k + z - 1));
  out.transition(K8 + k, F::sub(before_next, F::mul(full, row.loc(COL_UPDATED + k))));
  out.first(K9 + k, before);
}
}  // namespace keccak_sponge

// ------------------------------------------------------------------------------------------ AIR 7: multiplication
// One product per row: x * y = z + 2^256 w with x, y as sixteen 16-bit limbs and z, w by their bits.  Product column
// k (0 .. 31) collects the limb products x_i y_j with i + j = k and the carry of the column below:
//   is_mul * sum_{i+j=k} x_i y_j + c_(k-1) = p_k + 2^16 c_k,   p = the limbs of z (k < 16) then of w,   c_(-1) = 0 = c_31
// A column sum is below 16 * 2^32 + 2^21, so a carry fits 21 bits.  A row with is_mul = 0 is padding: the chain then
// forces every p and c to zero.
// Columns:
//   0            is_mul
//   1 .. 16      x limbs          17 .. 32   y limbs
//   33 .. 288    z bits: 33 + 16 k + j          289 .. 544   w bits: 289 + 16 k + j
//   545 .. 1216  carry bits: 545 + 21 k + j (carry out of product column k)
// Constraints (all rows):
//   U0  0            is_mul is a bit                                                       deg 2
//   U1  1 .. 256     z bits      U2  257 .. 512   w bits      U3  513 .. 1184   carry bits  deg 2
//   U4  1185 .. 1216 is_mul * sum_{i+j=k} x_i y_j + c_(k-1) - p_k - 2^16 c_k               deg 3
//   U5  1217         c_31 (the product has 512 bits)                                       deg 1
// Units: unit u = product columns 4u .. 4u + 3 (their p bits, carry bits and U4); unit 0 also U0, unit 7 also U5.
namespace arithmetic_mul {
constexpr uint32_t N_COLS = 1217, N_CONSTRAINTS = 1218, N_UNITS = 8;
constexpr uint32_t COL_MUL = 0, COL_X = 1, COL_Y = 17, COL_Z = 33, COL_W = 289, COL_CARRY = 545;
constexpr uint32_t U0 = 0, U1 = 1, U2 = 257, U3 = 513, U4 = 1185, U5 = 1217;
// sum_j 2^j bit_j over `n` bit columns starting at `col`, with their booleanity constraints starting at index `cidx`
template <class T, class Row, class Emit>
GL_HD T bits_value(const Row& row, Emit& out, uint32_t col, uint32_t n, uint32_t cidx) {
  typedef Ops<T> F;
  T v = F::k(0);
#pragma unroll 1
  for (uint32_t j = n; j > 0; j--) {
    const T b = row.loc(col + j - 1);
    out.all(cidx + j - 1, F::sub(F::mul(b, b), b));
    v = F::add(F::dbl(v), b);
  }
  return v;
}
template <class T, class Row, class Emit>
GL_HD void eval_unit(uint32_t u, const Row& row, Emit& out) {
  typedef Ops<T> F;
  const T m = row.loc(COL_MUL);
  if (u == 0) out.all(U0, F::sub(F::mul(m, m), m));
  // the carry into column 4u is the carry out of column 4u - 1 (its booleanity belongs to the unit below)
  T cin = F::k(0);
  if (u) {
#pragma unroll 1
    for (uint32_t j = 21; j > 0; j--) cin = F::add(F::dbl(cin), row.loc(COL_CARRY + 21 * (4 * u - 1) + j - 1));
  }
#pragma unroll 1
  for (uint32_t k = 4 * u; k < 4 * u + 4; k++) {
    const T p = k < 16 ? bits_value<T>(row, out, COL_Z + 16 * k, 16, U1 + 16 * k)
                       : bits_value<T>(row, out, COL_W + 16 * (k - 16), 16, U2 + 16 * (k - 16));
    const T c = bits_value<T>(row, out, COL_CARRY + 21 * k, 21, U3 + 21 * k);
    T conv = F::k(0);  // sum over i + j = k, 0 <= i, j < 16
    const uint32_t i0 = k < 16 ? 0 : k - 15, i1 = k < 16 ? k : 15;
#pragma unroll 1
    for (uint32_t i = i0; i <= i1; i++) conv = F::add(conv, F::mul(row.loc(COL_X + i), row.loc(COL_Y + k - i)));
    out.all(U4 + k, F::sub(F::add(F::mul(m, conv), cin), F::add(p, F::mul(F::k(65536), c))));
    cin = c;
  }
  if (u == 7) out.all(U5, cin);
}
}  // namespace arithmetic_mul

// ------------------------------------------------------------------------------------------ AIR 8: plonk
// A PLONK-shaped circuit as a table: what upstream's recursion circuits are proved with (CircuitData::prove, reached
// from proof_gen.rs:44-52 for the per-table shrink chains and from :66-75, :97-103 for aggregation and block proofs;
// CircuitConfig::standard_recursion_config [UPSTREAM-UNVERIFIED]: 135 wires of which 80 are routed, constants and
// sigmas preprocessed, quotient degree factor 8, two challenges).  NOT upstream's gate set and NOT a verifier circuit
// (a proof of this AIR does not check a child proof): it is the proof SYSTEM of the recursion layer -- gates selected
// by preprocessed constants, public inputs bound in-circuit, and the copy-constraint permutation argument with Z and
// partial products -- on a fixed satisfiable circuit.
// Wires (135 columns): routed 0..79 = 20 slots (a, b, c, d) = 4s .. 4s+3; advice 80..134 = 11 S-box units
// (x, x^2, x^4, x^6, x^7) = 80 + 5i ...
// Preprocessed constants (84 columns): 0 q_arith, 1 q_sbox (gate selectors), 2 c0, 3 c1 (gate constants),
// 4..83 sigma_j, j < 80: the copy permutation, sigma_j(w^i) = k_j' w^i' for the position (j', i') that wire (j, i) is
// tied to (k_j = 7^j: 80 disjoint cosets of the row subgroup).
// Gates (index, kind, degree):
//   G0  0 .. 19   all rows   q_arith (c0 a_s b_s + c1 c_s - d_s)                     (ArithmeticGate shape: 20 ops per row)  4
//   G1  20 .. 63  all rows   q_sbox (x2 - x x), (x4 - x2 x2), (x6 - x4 x2), (x7 - x6 x)  per unit (the Poseidon S-box x^7)   3
//   G2  64 .. 85  all rows   q_sbox (x - a_i), q_sbox (x7 - d_i): the unit's input and output are routed wires            2
//   G3  86 .. 89  first row  w_j - pub_j, j < 4: the public inputs (the hash of the proof's public-input list)             1
// The copy constraints are the table's auxiliary columns (namespace ctl below): per challenge set Z and nine partial
// products.  The fixed circuit (rows in groups of four after a first group that holds the public-input row and no-ops):
//   row 4g      arithmetic, inputs free             d_s = c0 a_s b_s + c1 c_s
//   row 4g + 1  arithmetic, a_s := d_s(4g), b_s := d_(s+1)(4g), c_s := c_s(4g)          (3- and 2-cycles)
//   row 4g + 2  S-box, x_i := d_i(4g + 1) through a_i, d_i = x_i^7                       (2-cycles)
//   row 4g + 3  arithmetic, a_i := d_i(4g + 2) for i < 11, the rest free               (2-cycles)
// and c_j(first arithmetic row) := pub_j (row 0), j < 4, ties the computation to the public inputs.
//
// Round 5: the circuit HASHES its public-input list itself (upstream: the public inputs of a recursion circuit are hashed
// in-circuit by PoseidonGates and the four hash words routed to the PublicInputGate; round 4 bound the first row to a hash
// the HOST computed).  A **Poseidon gate** -- one whole permutation per row across the wires, as upstream's PoseidonGate
// lays it out, which is why a row has 135 wires -- selected by a fifth constant column q_hash:
//   wires of a hash row: 0..11 the state in, 12..23 the state out, 24..59 the S-box inputs of full rounds 1..3 (round 0's
//   are in + rc_0), 60..81 the S-box input (word 0) of the 22 partial rounds, 82..129 those of full rounds 26..29
//   G4  90 .. 207  all rows   q_hash (wire - what the round function makes of the previous wires)   118 constraints,  degree 8
//        90 + 12 (r - 1) + i   s_r[i] - (M (s_(r-1))^7 + rc_r)[i],  r = 1..3          (s_0 = in + rc_0)
//        126 + (r - 4)         p_r - state[0] through the partial rounds r = 4..25        (state: M applied to (p^7, the other
//                                                                                        eleven words), + rc_(r+1))
//        148 + 12 (r - 26) + i s_r[i] - state[i],  r = 26..29;    196 + i   out[i] - (M (s_29)^7)[i]
//   G5  208 .. 212 all rows   q_hash s (s - 1),  q_hash (delta_i - s (in[4 + i] - in[i])), i < 4:  the gate's swap -- wire 130
//        = s, wires 131..134 = delta; the permutation runs on (in[i] + delta_i, in[4 + i] - delta_i, in[8..11]), i.e. on
//        (left, right, capacity) of a Merkle step whose node arrives in in[0..3] and whose sibling in in[4..7]       degree 3
// with the permutation's own round constants and MDS matrix (poseidon_rc.inc): hash rows compute hash_no_pad (s = 0).
// All 135 wires of a Poseidon row are now spoken for, as in upstream's PoseidonGate (12 + 12 + 1 + 4 + 36 + 22 + 48).
// Rows 4 .. 4 + H - 1 (H = ceil(len / 8) <= 13) absorb the list eight words at a time: the words are free wires 0..7 of
// their row, the other state words are copies -- of the previous row's output (the sponge's carry: words 8..11, and the
// words a short last chunk leaves alone), or, in the first row, of ZERO wires: row 1 is an arithmetic row with
// c0 = c1 = 0, so its twenty d wires are zero.  The first four output words of the last hash row ARE the public-input
// wires of row 0 (one copy cycle with c_j of the first two arithmetic rows), which G3 binds to the four words the verifier computed from
// the list it was given.  Rows 4..16 are the hash region (no-ops beyond H <= 13).
// **Merkle rows** (rows 17 .. 17 + n_paths x depth - 1, also selected by q_hash): a recursion circuit walks,
// per child proof, the Merkle path of the child's first query into its trace oracle (merkle_proofs::
// verify_merkle_proof_to_cap in-circuit): level l's row takes the node of level l in in[0..3] -- a copy of the row
// below's out[0..3]; at l = 0 a copy of the leaf-digest words of the public-input list -- the sibling in in[4..7] (free
// witness, like the position bit on the swap wire) and zeros in in[8..11] (one copy cycle through all of them and zero
// wire 19 of row 1); out[0..3] of the path's last row is a copy of the list's cap-entry words.  So the hash the verifier
// is given commits to (leaf digest, cap entry) pairs between which the circuit has checked a path; which leaf and which
// cap the words are is the aggregating host's statement, as the child digests in the same list are.  Arithmetic groups
// start at the first multiple of four past the Merkle rows (row 20 for a circuit that walks no path).
// **Leaf rows** (Layout::leaf_len > 0; right after the Merkle rows, ceil(leaf_len / 8) per path): the circuit hashes the
// row each path starts from -- a second sponge whose words are free wires -- and its last output IS the path's first node.
namespace plonk {
constexpr uint32_t N_COLS = 135, N_CONST = 85, N_ROUTED = 80, N_SLOTS = 20, N_SBOX = 11, N_CONSTRAINTS = 213, N_UNITS = 11;
constexpr uint32_t CST_ARITH = 0, CST_SBOX = 1, CST_C0 = 2, CST_C1 = 3, CST_HASH = 4, CST_SIGMA = 5;
constexpr uint32_t COL_SBOX = 80;
constexpr uint32_t G0 = 0, G1 = 20, G2 = 64, G3 = 86, G4 = 90, G5 = 208;
constexpr uint32_t HASH_ROW0 = 4, HASH_ROWS_MAX = 13, ZERO_ROW = 1, MAX_PI = 8 * HASH_ROWS_MAX;
constexpr uint32_t MERKLE_ROW0 = HASH_ROW0 + HASH_ROWS_MAX, MERKLE_ROWS_MAX = 96, MERKLE_ZERO_COL = 79, LEAF_ROWS_MAX = 120;
constexpr uint32_t H_IN = 0, H_OUT = 12, H_FULL1 = 24, H_PART = 60, H_FULL2 = 82, H_SWAP = 130, H_DELTA = 131, H_WIRES = 135;
// What a circuit of this family does besides its arithmetic groups: it hashes a public-input list of pi_len words
// (rows 4 ..), and it walks n_paths Merkle paths of `depth` levels each (rows 12 ..): path p's leaf digest is list words
// path_pi0 + 8p .. + 3, the cap entry it must arrive at is list words path_pi0 + 8p + 4 .. + 7.
// leaf_len > 0: the circuit also HASHES, per path, the leaf the path starts from -- a row of leaf_len words (the child's
// opened trace row: free wires, not words of the list) absorbed eight at a time in rows of its own right after the
// Merkle rows; the digest it arrives at is the path's first node and the list's leaf-digest words (one copy cycle).
struct Layout {
  uint32_t pi_len, n_paths, depth, path_pi0, leaf_len = 0;
};
GL_HD uint32_t hash_rows(uint32_t pi_len) { return (pi_len + 7) / 8; }
GL_HD uint32_t merkle_rows(const Layout& L) { return L.n_paths * L.depth; }
GL_HD uint32_t leaf_rows(const Layout& L) { return L.leaf_len ? L.n_paths * hash_rows(L.leaf_len) : 0; }
GL_HD uint32_t leaf_row0(const Layout& L) { return MERKLE_ROW0 + merkle_rows(L); }
GL_HD uint32_t arith_row0(const Layout& L) { return (MERKLE_ROW0 + merkle_rows(L) + leaf_rows(L) + 3) & ~3u; }  // 20 without paths
GL_HD bool layout_ok(const Layout& L, uint32_t n) {
  return L.pi_len >= 1 && L.pi_len <= MAX_PI && merkle_rows(L) <= MERKLE_ROWS_MAX && (L.n_paths == 0 || L.depth >= 1) &&
         L.path_pi0 + 8 * L.n_paths <= L.pi_len && (L.leaf_len == 0 || (L.leaf_len > 8 && L.n_paths >= 1)) &&
         leaf_rows(L) <= LEAF_ROWS_MAX && arith_row0(L) + 4 <= n;
}
// is state word k of hash row h (0-based) a carried word -- a copy of the previous row's output (h > 0) or of a zero wire
// (h = 0) -- rather than a word of the list?
GL_HD bool hash_word_is_carried(uint32_t pi_len, uint32_t h, uint32_t k) {
  const uint32_t H = hash_rows(pi_len), last = pi_len - 8 * (H - 1);  // words in the last chunk: 1..8
  return k >= 8 || (h + 1 == H && k >= last);
}
// where wire (col, row) points in the copy permutation (the next position of its cycle), n rows
GL_HD void sigma_of(uint32_t col, uint32_t row, uint32_t n, const Layout& L, uint32_t& col_out, uint32_t& row_out) {
  col_out = col; row_out = row;
  if (col >= N_ROUTED) return;
  const uint32_t pi_len = L.pi_len, H = hash_rows(pi_len), last_hash_row = HASH_ROW0 + H - 1;
  const uint32_t A0 = arith_row0(L), T = merkle_rows(L);
  if (row < A0) {
    if (row == 0) {  // the public-input row: w_j(0) -> c_j(A0) -> c_j(A0 + 1) -> out_j(last hash row) -> w_j(0)
      if (col < 4) { col_out = 4 * col + 2; row_out = A0; }
    } else if (row == ZERO_ROW) {  // zero wire z = d_z(1) <-> the z-th carried word of the first hash row
      if (T && col == MERKLE_ZERO_COL) {  // the last zero wire heads the chain of the Merkle rows' capacity words
        col_out = H_IN + 8; row_out = MERKLE_ROW0;
      } else if ((col & 3) == 3) {
        uint32_t z = col >> 2, seen = 0;
        for (uint32_t k = 0; k < 12; k++)
          if (hash_word_is_carried(pi_len, 0, k)) {
            if (seen == z) { col_out = H_IN + k; row_out = HASH_ROW0; }
            seen++;
          }
      }
    } else if (row >= HASH_ROW0 && row <= last_hash_row) {
      const uint32_t h = row - HASH_ROW0;
      if (col < 12) {  // a state word coming in
        if (hash_word_is_carried(pi_len, h, col)) {
          if (h == 0) {
            uint32_t z = 0;
            for (uint32_t k = 0; k < col; k++) z += hash_word_is_carried(pi_len, 0, k);
            col_out = 4 * z + 3; row_out = ZERO_ROW;
          } else {
            col_out = H_OUT + col; row_out = row - 1;
          }
        } else {  // a word of the list: a path's leaf digest / cap entry is the same wire as the path's first input / last output
          const uint32_t i = 8 * h + col;
          if (i >= L.path_pi0 && i < L.path_pi0 + 8 * L.n_paths) {
            const uint32_t q = i - L.path_pi0, p = q >> 3, r = q & 7;
            if (r < 4) { col_out = H_IN + r; row_out = MERKLE_ROW0 + p * L.depth; }
            else { col_out = H_OUT + r - 4; row_out = MERKLE_ROW0 + p * L.depth + L.depth - 1; }
          }
        }
      } else if (col < 24) {  // a state word going out
        const uint32_t k = col - H_OUT;
        if (row == last_hash_row) {
          if (k < 4) { col_out = k; row_out = 0; }
        } else if (hash_word_is_carried(pi_len, h + 1, k)) {
          col_out = H_IN + k; row_out = row + 1;
        }
      }
    } else if (row >= MERKLE_ROW0 && row < MERKLE_ROW0 + T) {
      // level l of path p: in 0..3 the node below (the leaf digest at l = 0), 4..7 its sibling (free), 8..11 zero;
      // out 0..3 the node above (the cap entry at the last level)
      const uint32_t t = row - MERKLE_ROW0, p = t / L.depth, l = t % L.depth;
      if (col < 4) {
        if (l == 0 && L.leaf_len) {  // the leaf digest: computed by the path's leaf rows (their last output), which closes the cycle
          col_out = H_OUT + col; row_out = leaf_row0(L) + p * hash_rows(L.leaf_len) + hash_rows(L.leaf_len) - 1;
        } else if (l == 0) { const uint32_t i = L.path_pi0 + 8 * p + col; col_out = i & 7; row_out = HASH_ROW0 + (i >> 3); }
        else { col_out = H_OUT + col; row_out = row - 1; }
      } else if (col >= 8 && col < 12) {  // one cycle through every capacity word of every Merkle row and a zero wire
        if (col < 11) col_out = col + 1;
        else if (t + 1 < T) { col_out = 8; row_out = row + 1; }
        else if (L.leaf_len) { col_out = 8; row_out = leaf_row0(L); }  // ... and on through the leaf sponges' first rows
        else { col_out = MERKLE_ZERO_COL; row_out = ZERO_ROW; }
      } else if (col >= H_OUT && col < H_OUT + 4) {
        const uint32_t j = col - H_OUT;
        if (l + 1 == L.depth) { const uint32_t i = L.path_pi0 + 8 * p + 4 + j; col_out = i & 7; row_out = HASH_ROW0 + (i >> 3); }
        else { col_out = H_IN + j; row_out = row + 1; }
      }
    } else if (L.leaf_len && row >= leaf_row0(L) && row < leaf_row0(L) + leaf_rows(L)) {
      // row h of the leaf sponge of path p: eight words of the leaf (free), the rest carried; the first row's capacity
      // words are zeros (members of the Merkle rows' zero cycle); the last output is the path's first node
      const uint32_t LH = hash_rows(L.leaf_len), t = row - leaf_row0(L), p = t / LH, h = t % LH;
      if (col < 12) {
        if (hash_word_is_carried(L.leaf_len, h, col)) {
          if (h > 0) { col_out = H_OUT + col; row_out = row - 1; }
          else if (col < 11) col_out = col + 1;                                  // (leaf_len > 8: the carried words of row 0 are 8..11)
          else if (p + 1 < L.n_paths) { col_out = 8; row_out = row + LH; }
          else { col_out = MERKLE_ZERO_COL; row_out = ZERO_ROW; }
        }
      } else if (col < 24) {
        const uint32_t k = col - H_OUT;
        if (h + 1 == LH) {
          if (k < 4) { const uint32_t i = L.path_pi0 + 8 * p + k; col_out = i & 7; row_out = HASH_ROW0 + (i >> 3); }
        } else if (hash_word_is_carried(L.leaf_len, h + 1, k)) {
          col_out = H_IN + k; row_out = row + 1;
        }
      }
    }
    return;
  }
  const uint32_t g = row >> 2, p = row & 3, s = col >> 2, w = col & 3;
  const uint32_t base = 4 * g;
  if (p == 0) {
    if (w == 3) { col_out = 4 * s; row_out = base + 1; }                        // d_s(4g) -> a_s(4g+1)
    else if (w == 2) { col_out = col; row_out = base + 1; }                     // c_s(4g) -> c_s(4g+1)
  } else if (p == 1) {
    if (w == 0) { col_out = 4 * ((s + N_SLOTS - 1) % N_SLOTS) + 1; row_out = base + 1; }  // a_s -> b_(s-1) of the same row
    else if (w == 1) { col_out = 4 * ((s + 1) % N_SLOTS) + 3; row_out = base; }           // b_s(4g+1) -> d_(s+1)(4g)
    else if (w == 2) {                                                                   // c_s(4g+1) -> c_s(4g), or on to the hash output
      if (base == A0 && s < 4) { col_out = H_OUT + s; row_out = last_hash_row; }
      else { col_out = col; row_out = base; }
    } else if (s < N_SBOX) { col_out = 4 * s; row_out = base + 2; }             // d_i(4g+1) -> a_i(4g+2)
  } else if (p == 2) {
    if (s < N_SBOX && w == 0) { col_out = 4 * s + 3; row_out = base + 1; }      // a_i(4g+2) -> d_i(4g+1)
    else if (s < N_SBOX && w == 3) { col_out = 4 * s; row_out = base + 3; }     // d_i(4g+2) -> a_i(4g+3)
  } else {
    if (s < N_SBOX && w == 0) { col_out = 4 * s + 3; row_out = base + 2; }      // a_i(4g+3) -> d_i(4g+2)
  }
}
// 7^(8u): the coset shift k_j = 7^j of the first routed wire of chunk u
GL_HD uint64_t chunk_shift(uint32_t u) {
  constexpr uint64_t K8[10] = {0x1ULL, 0x57f6c1ULL, 0x1e39a5057d81ULL, 0x62b942f056949437ULL, 0x33cdc3006affab59ULL,
                               0xcb893e19494e73dULL, 0x36f01f6fb02d9400ULL, 0xd844c68aea81b372ULL, 0x9c232dfb69ab851eULL,
                               0x255bd1d3b936892aULL};
  return K8[u];
}
// Unit u (0..9) = everything that reads routed wires 8u .. 8u+7 (slots 2u, 2u+1), each wire and each sigma read ONCE:
//   * the arithmetic gates of the two slots (G0);
//   * chunk u + 1 of the copy constraints for BOTH challenge sets (the statement above ctl::eval, index ctl_base + 11 c + u:
//     cur den - prev num with prev = Z(next row) for the first chunk, cur = Z(this row) for the last), and for u = 9 the
//     first-row constraints Z_c - 1;
//   * the S-box units whose wires lie in the chunk (2u and 2u + 1 for u < 5, unit 10 for u = 5) (G1, G2);
//   * for u = 0 the public-input row (G3).
// (Round 4 had ONE gate unit and the copy constraints as a second pass per challenge set: every routed wire was read
// three times and every sigma twice -- 2.05 x the algorithmic bytes by PMC --, and a lone proof's 2^16 rows were 256
// workgroups with nothing to spread.)  Every constraint keeps its index, so the quotient is the same polynomial.
// Unit 10: the Poseidon gate (G4).  The wires of the row are taken at their word: every constraint compares a wire with
// what the round function makes of the wires before it, so a hash row is a chain of 118 local checks (+ 5 for the swap).
template <class T, class Row, class Emit, bool SELECTOR = true>
GL_HD void eval_hash_unit(const Row& row, Emit& out) {
  // SELECTOR = false: the differences go out bare and the caller multiplies the folded sum by q_hash once
  // (sum_i alpha^e_i q d_i = q sum_i alpha^e_i d_i: the device's hash pass, which folds nothing else)
  typedef Ops<T> F;
  typedef PoseidonOps<T> H;
  const T qh = row.cst(CST_HASH);
  const T q4[4] = {qh, qh, qh, qh};
  T st[12];
  {
    // the swap: with s = wire 130 (a bit) and delta_i = s (in[4 + i] - in[i]) (wires 131..134), the permutation takes
    // (in[i] + delta_i, in[4 + i] - delta_i, in[8..11]): the two digests of a Merkle step in the order the path bit says
    const T sw = row.loc(H_SWAP);
    T lhs[4], rhs[4], dl[4], pr[4], g[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      lhs[i] = row.loc(H_IN + i);
      rhs[i] = row.loc(H_IN + 4 + i);
      dl[i] = row.loc(H_DELTA + i);
      pr[i] = F::sub(rhs[i], lhs[i]);
    }
    const T s4[4] = {sw, sw, sw, sw};
    F::mul4(s4, pr, g);
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      const T d = F::sub(dl[i], g[i]);
      out.all(G5 + 1 + i, SELECTOR ? F::mul(qh, d) : d);
      st[i] = H::add_rc0(F::add(lhs[i], dl[i]), i);
      st[4 + i] = H::add_rc0(F::sub(rhs[i], dl[i]), 4 + i);
      st[8 + i] = H::add_rc0(row.loc(H_IN + 8 + i), 8 + i);
    }
    const T bit = F::mul(sw, F::sub(sw, F::k(1)));
    out.all(G5, SELECTOR ? F::mul(qh, bit) : bit);
  }
  // a block of twelve wires against the state, selector applied four at a time; the wires become the state
  auto check12 = [&](uint32_t col0, uint32_t idx0) {
#pragma unroll
    for (uint32_t i0 = 0; i0 < 12; i0 += 4) {
      T d[4], g[4];
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) {
        const T wv = row.loc(col0 + i0 + i);
        d[i] = F::sub(wv, st[i0 + i]);
        st[i0 + i] = wv;
      }
      if (SELECTOR) F::mul4(q4, d, g);
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) out.all(idx0 + i0 + i, SELECTOR ? g[i] : d[i]);
    }
  };
#pragma unroll 1
  for (uint32_t r = 1; r <= 3; r++) {
    H::sbox_all(st);
    H::mds_rc(st, (int)r);
    check12(H_FULL1 + 12 * (r - 1), G4 + 12 * (r - 1));
  }
  H::sbox_all(st);
  H::mds_rc_word0(st, 4);
#pragma unroll 1
  for (uint32_t r = 4; r <= 25; r++) {
    const T pv = row.loc(H_PART + r - 4);
    const T d = F::sub(pv, st[0]);
    out.all(G4 + 36 + (r - 4), SELECTOR ? F::mul(qh, d) : d);
    st[0] = H::sbox(pv);
    if (r < 25) H::mds_rc_word0(st, (int)r + 1);
    else H::mds_rc(st, (int)r + 1);
  }
#pragma unroll 1
  for (uint32_t r = 26; r <= 29; r++) {
    check12(H_FULL2 + 12 * (r - 26), G4 + 58 + 12 * (r - 26));
    H::sbox_all(st);
    H::mds_rc(st, r < 29 ? (int)r + 1 : -1);
  }
  check12(H_OUT, G4 + 106);
}
// Units 0 .. 9: the gates of slots 2u, 2u + 1, the copy constraints of chunk u + 1, the S-box units and the public row.
template <class T, class Row, class Emit>
GL_HD void eval_chunk_unit(uint32_t u, uint32_t ctl_base, const uint64_t ctl[4], const Row& row, Emit& out) {
  typedef Ops<T> F;
  T w[8], sg[8];
#pragma unroll
  for (uint32_t i = 0; i < 8; i++) w[i] = row.loc(8 * u + i);
#pragma unroll
  for (uint32_t i = 0; i < 8; i++) sg[i] = row.cst(CST_SIGMA + 8 * u + i);
  const T qa = row.cst(CST_ARITH), qs = row.cst(CST_SBOX), c0 = row.cst(CST_C0), c1 = row.cst(CST_C1);
  const T x = row.x(), kp = F::k(chunk_shift(u));
  const T beta[2] = {F::k(ctl[0]), F::k(ctl[2])}, gamma[2] = {F::k(ctl[1]), F::k(ctl[3])};
  // ---- gates: d - c0 a b - c1 c, times the selector; the spare slots of the groups start the two beta x k chains
  T bkx[2];
  {
    const T l1[4] = {w[0], w[4], c1, c1}, r1[4] = {w[1], w[5], w[2], w[6]};
    T p1[4];
    F::mul4(l1, r1, p1);  // a b of the two slots, c1 c of the two slots
    const T l2[4] = {c0, c0, beta[0], beta[1]}, r2[4] = {p1[0], p1[1], x, x};
    T p2[4];
    F::mul4(l2, r2, p2);  // c0 a b; beta_c x
    const T t0 = F::sub(F::add(p2[0], p1[2]), w[3]), t1 = F::sub(F::add(p2[1], p1[3]), w[7]);
    const T l3[4] = {qa, qa, p2[2], p2[3]}, r3[4] = {t0, t1, kp, kp};
    T p3[4];
    F::mul4(l3, r3, p3);
    out.all(G0 + 2 * u, p3[0]);
    out.all(G0 + 2 * u + 1, p3[1]);
    bkx[0] = p3[2];
    bkx[1] = p3[3];
  }
  // ---- copy constraints of chunk k = u + 1, both challenge sets from the same wires and sigmas
  T half[2][4];  // per set: numerator halves A, B, denominator halves A, B
#pragma unroll
  for (uint32_t c = 0; c < 2; c++) {
    const T b4[4] = {beta[c], beta[c], beta[c], beta[c]};
    T pn[4], pd[4];
#pragma unroll
    for (uint32_t h = 0; h < 2; h++) {
      const T s4[4] = {sg[4 * h], sg[4 * h + 1], sg[4 * h + 2], sg[4 * h + 3]};
      T bs[4], tn[4], td[4];
      F::mul4_lazy(b4, s4, bs);
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) {
        const T wg = F::add(w[4 * h + i], gamma[c]);
        tn[i] = F::add_lazy(bkx[c], wg);
        td[i] = F::add_lazy(bs[i], wg);
        bkx[c] = F::mul7(bkx[c]);
      }
      const T l4[4] = {tn[0], tn[2], td[0], td[2]}, r4[4] = {tn[1], tn[3], td[1], td[3]};
      T q4[4];
      F::mul4_lazy(l4, r4, q4);
      pn[2 * h] = q4[0]; pn[2 * h + 1] = q4[1]; pd[2 * h] = q4[2]; pd[2 * h + 1] = q4[3];
    }
    const T l4[4] = {pn[0], pn[2], pd[0], pd[2]}, r4[4] = {pn[1], pn[3], pd[1], pd[3]};
    F::mul4_lazy(l4, r4, half[c]);
  }
  {
    const T l4[4] = {half[0][0], half[0][2], half[1][0], half[1][2]}, r4[4] = {half[0][1], half[0][3], half[1][1], half[1][3]};
    T nd[4];  // num_0, den_0, num_1, den_1
    F::mul4_lazy(l4, r4, nd);
    const uint32_t k = u + 1;
    T cur[2], prev[2];
#pragma unroll
    for (uint32_t c = 0; c < 2; c++) {
      cur[c] = row.aux(10 * c + (k < 10 ? k : 0));
      prev[c] = k == 1 ? row.aux_nxt(10 * c) : row.aux(10 * c + k - 1);
    }
    const T m4[4] = {cur[0], prev[0], cur[1], prev[1]}, n4[4] = {nd[1], nd[0], nd[3], nd[2]};
    T g4[4];
    F::mul4(m4, n4, g4);
    out.all(ctl_base + u, F::sub(g4[0], g4[1]));
    out.all(ctl_base + 11 + u, F::sub(g4[2], g4[3]));
    if (u == 9) {
      out.first(ctl_base + 10, F::sub(cur[0], F::k(1)));
      out.first(ctl_base + 21, F::sub(cur[1], F::k(1)));
    }
  }
  // ---- the S-box units of the chunk: x^2, x^4, x^6, x^7 relations and the two wire relations, times the selector
  if (u < 6) {
    const uint32_t n_here = u < 5 ? 2 : 1;
    const T qs4[4] = {qs, qs, qs, qs};
    T pend[4] = {F::k(0), F::k(0), F::k(0), F::k(0)};
#pragma unroll
    for (uint32_t e = 0; e < 2; e++) {
      if (e >= n_here) break;
      const uint32_t i = 2 * u + e, col = COL_SBOX + 5 * i;
      const T xs = row.loc(col), x2 = row.loc(col + 1), x4 = row.loc(col + 2), x6 = row.loc(col + 3), x7 = row.loc(col + 4);
      const T l4[4] = {xs, x2, x4, x6}, r4[4] = {xs, x2, x2, xs};
      T p4[4], d4[4], g4[4];
      F::mul4(l4, r4, p4);
      d4[0] = F::sub(x2, p4[0]); d4[1] = F::sub(x4, p4[1]); d4[2] = F::sub(x6, p4[2]); d4[3] = F::sub(x7, p4[3]);
      F::mul4(qs4, d4, g4);
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) out.all(G1 + 4 * i + q, g4[q]);
      pend[2 * e] = F::sub(xs, w[4 * e]);          // wire a of slot i = 8u + 4e
      pend[2 * e + 1] = F::sub(x7, w[4 * e + 3]);  // wire d of slot i
    }
    T w4[4];
    F::mul4(qs4, pend, w4);
    out.all(G2 + 4 * u, w4[0]);
    out.all(G2 + 4 * u + 1, w4[1]);
    if (n_here == 2) {
      out.all(G2 + 4 * u + 2, w4[2]);
      out.all(G2 + 4 * u + 3, w4[3]);
    }
  }
  if (u == 0) {
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) out.first(G3 + j, F::sub(w[j], F::k(row.pub(j))));
  }
}
template <class T, class Row, class Emit>
GL_HD void eval_unit(uint32_t u, uint32_t ctl_base, const uint64_t ctl[4], const Row& row, Emit& out) {
  if (u == 10) eval_hash_unit<T>(row, out);
  else eval_chunk_unit<T>(u, ctl_base, ctl, row, out);
}
}  // namespace plonk

// ------------------------------------------------------------------------------------------ registry
GL_HD uint32_t n_constraints(const Shape& s) {
  return s.air_id == KECCAK_F ? keccak::N_CONSTRAINTS
         : s.air_id == LOGIC  ? logic::N_CONSTRAINTS
         : s.air_id == MEMORY ? memory::N_CONSTRAINTS
         : s.air_id == ARITHMETIC ? arithmetic::N_CONSTRAINTS
         : s.air_id == BYTE_PACKING ? byte_packing::N_CONSTRAINTS
         : s.air_id == KECCAK_SPONGE ? keccak_sponge::N_CONSTRAINTS
         : s.air_id == ARITHMETIC_MUL ? arithmetic_mul::N_CONSTRAINTS
         : s.air_id == PLONK ? plonk::N_CONSTRAINTS
                              : synthetic::n_constraints(s);
}
GL_HD uint32_t n_units(const Shape& s) {
  return s.air_id == KECCAK_F ? keccak::N_UNITS
         : s.air_id == LOGIC  ? logic::N_UNITS
         : s.air_id == MEMORY ? memory::N_UNITS
         : s.air_id == ARITHMETIC ? arithmetic::N_UNITS
         : s.air_id == BYTE_PACKING ? byte_packing::N_UNITS
         : s.air_id == KECCAK_SPONGE ? keccak_sponge::N_UNITS
         : s.air_id == ARITHMETIC_MUL ? arithmetic_mul::N_UNITS
         : s.air_id == PLONK ? plonk::N_UNITS
                              : synthetic::n_units(s);
}
// ctl_base / ctl: the index of the table's first lookup constraint and the lookup challenges -- AIR 8's units carry its
// copy constraints (the other tables' lookup constraints are ctl::eval's)
template <class T, class Row, class Emit>
GL_HD void eval_unit(const Shape& s, uint32_t unit, uint32_t ctl_base, const uint64_t ctl[4], const Row& row, Emit& out) {
  if (s.air_id == KECCAK_F) keccak::eval_unit<T>(unit, row, out);
  else if (s.air_id == LOGIC) logic::eval_unit<T>(unit, ctl_base, ctl, row, out);
  else if (s.air_id == MEMORY) memory::eval_unit<T>(row, out);
  else if (s.air_id == ARITHMETIC) arithmetic::eval_unit<T>(unit, row, out);
  else if (s.air_id == BYTE_PACKING) byte_packing::eval_unit<T>(unit, row, out);
  else if (s.air_id == KECCAK_SPONGE) keccak_sponge::eval_unit<T>(unit, ctl_base, ctl, row, out);
  else if (s.air_id == ARITHMETIC_MUL) arithmetic_mul::eval_unit<T>(unit, row, out);
  else if (s.air_id == PLONK) plonk::eval_unit<T>(unit, ctl_base, ctl, row, out);
  else synthetic::eval_unit<T>(s, unit, row, out);
}

// ------------------------------------------------------------------------------------------ cross-table lookups
// The auxiliary columns of a table (committed after the four lookup challenges beta_0, gamma_0, beta_1, gamma_1 are
// known; upstream: plonky2_evm's cross_table_lookup, whose Z columns sit beside Stark::eval -- reached from
// proof_gen.rs:44-52, where ONE call proves the seven tables of prover_state.rs:85-93 as one statement).  Their
// constraints follow the AIR's in the constraint list (index base = the AIR's count).
//
// A lookup is a filtered running product.  With a filter f (0 / 1 on every row) and a tuple of column values
// compressed by a challenge set c = (beta_c, gamma_c), v_c = sum_j beta_c^j t_j,
//     z_c[i] = prod_{i' >= i} term_c[i'],   term_c = 1 + f (gamma_c + v_c - 1)
// (rows the filter leaves out contribute a factor 1).  Constraints per product column:  transition  z - z' term,
// last row  z - term.  z_c opened at the first row (every aux column is, proof layout "open_first") is the product
// over the table's filtered tuples; the statement "the tuples the looking tables send are the tuples the looked table
// exposes" is the equality of those first-row values (ctl::pairs() below), checked for both challenge sets by whoever
// verifies a transaction's table proofs (bp_verify_txn_table_proofs; upstream's root circuit does it in-circuit).
// Every lookup exists twice, once per challenge set (upstream repeats its CTLs num_challenges = 2 times).
//
// AIR 0  synthetic   n_cols / 8 columns: the running product over trace columns 8k, 8k+1 with challenge set k mod 2,
//                    no filter.  A load placeholder from before the real tables existed:
//                    nothing looks these products up.
// AIR 1  keccak_f    4 columns: h_0 h_1 | z_0 z_1.  LOOKED table of "keccak_sponge -> keccak_f": the tuple is a
//                    whole permutation (50 input limbs, 50 output limbs).  Input and output live 23 rows apart, so
//                    h_c carries the compressed input along the permutation's rows -- s_0 (h_c - sum_j beta_c^j A_j) = 0
//                    on every row, (1 - s_23)(h_c' - h_c) = 0 on transitions -- and the tuple is read on the last
//                    round's row: v_c = h_c + beta_c^50 sum_j beta_c^j out_j, out = the iota output for lane 0, else
//                    the chi output.  g is the filter: which permutations the table exposes (g (g - 1) = 0,
//                    g (1 - s_23) = 0: only last-round rows).  A table holds padding permutations nobody asks for;
//                    exposing a subset is sound because every row of the table is a valid permutation by the AIR and
//                    g is a TRACE column (2430), committed before the challenges.
// AIR 6  keccak_sponge  2 columns: z_0 z_1.  LOOKING side of the same lookup: filter is_full + is_final (every row
//                    that absorbs a block), tuple = (xored rate limbs, capacity limbs | updated state limbs).
// AIR 5  byte_packing  2 columns: z_0 z_1.  LOOKING side of "byte_packing -> memory": filter = the row has a length,
//                    tuple = (is_read, address, timestamp, the eight value limbs): the word the sequence spells is ONE
//                    operation of the memory table, whose rows hold 256-bit values.  (Upstream looks every BYTE of the
//                    sequence up, at consecutive addresses; here the word is the unit, as in this memory table.)
// AIR 3  memory      2 columns: z_0 z_1.  LOOKED side: g (TRACE column 44) is the filter, which operations the table
//                    exposes (g (g - 1) = 0); tuple = (is_read, address, timestamp, value limbs) of the row.
// AIR 6  keccak_sponge  ... and 10 more columns, LOOKING side of "keccak_sponge -> logic" (upstream: ctl_logic's
//                    keccak_sponge_stark::ctl_looking_logic(i), i < 5 [UPSTREAM-UNVERIFIED]): the XOR of the rate with
//                    the block, eight 32-bit limbs at a time (136 bytes = 34 limbs: five operations, the last one two
//                    limbs and six zeros), is an operation of the logic table: column 2 + 2 m + c, tuple (is_and, is_or,
//                    is_xor = 1 | rate-before limbs | block limbs | xored limbs), the inputs as sums over their bit columns.
// AIR 2  logic       2 columns: z_0 z_1.  LOOKED side: g (TRACE column 523) is the filter, which operations the table
//                    exposes; tuple = the row's flags, the limbs of its inputs (sums over the bit columns) and of its result.
//                    The product of the sponge table's five first-row values equals z_c's.
// AIR 4, 7           1 column: no lookup is built for these tables (upstream's go through the CPU table, which needs
//                    the EVM interpreter): a constant running product z = 1 keeps the oracle set of every table the same.
namespace ctl {
// The FILTER columns of the three looked tables (which rows are exposed) are TRACE columns (keccak::COL_G, memory::COL_G, logic::COL_G):
// they are committed before the lookup challenges (beta, gamma) are drawn.  (Round 4 kept them among the auxiliary
// columns, committed AFTER the challenges: a prover could then pick the exposed subset knowing the challenges -- a
// subset-product search over a smooth multiplicative group)  Upstream keeps its filters in the trace too.
constexpr uint32_t KECCAK_H = 0, KECCAK_Z = 2, KECCAK_N_AUX = 4, KECCAK_N_CONSTRAINTS = 10;
// sponge: z_0 z_1 (-> keccak_f), then per block-of-eight-limbs m < 5 and challenge set c the product 2 + 2 m + c (-> logic)
constexpr uint32_t SPONGE_Z = 0, SPONGE_LZ = 2, SPONGE_LOGIC_OPS = 5, SPONGE_N_AUX = 2 + 2 * SPONGE_LOGIC_OPS, SPONGE_N_CONSTRAINTS = 2 * SPONGE_N_AUX;
constexpr uint32_t LOGIC_Z = 0, LOGIC_N_AUX = 2, LOGIC_N_CONSTRAINTS = 5;
constexpr uint32_t LOGIC_TUPLE = 27;  // is_and, is_or, is_xor, eight limbs of each input, eight of the result
constexpr uint32_t TUPLE_LIMBS = 50;  // a Keccak state as 32-bit limbs
constexpr uint32_t PACK_Z = 0, PACK_N_AUX = 2, PACK_N_CONSTRAINTS = 4;
constexpr uint32_t MEM_Z = 0, MEM_N_AUX = 2, MEM_N_CONSTRAINTS = 5;
constexpr uint32_t WORD_TUPLE = 11;  // is_read, address, timestamp, eight value limbs
// AIR 8 (plonk): per challenge set c, column 10 c = Z_c and 10 c + k = the k-th partial product, k = 1..9
constexpr uint32_t PLONK_N_AUX = 20, PLONK_N_CONSTRAINTS = 22, PLONK_CHUNK = 8, PLONK_CHUNKS = 10;

GL_HD uint32_t n_aux(const Shape& s) {
  return s.air_id == SYNTHETIC ? s.n_cols / 8
         : s.air_id == KECCAK_F ? KECCAK_N_AUX
         : s.air_id == KECCAK_SPONGE ? SPONGE_N_AUX
         : s.air_id == BYTE_PACKING ? PACK_N_AUX
         : s.air_id == MEMORY ? MEM_N_AUX
         : s.air_id == LOGIC ? LOGIC_N_AUX
         : s.air_id == PLONK ? PLONK_N_AUX
                             : 1;
}
GL_HD uint32_t n_constraints(const Shape& s) {
  return s.air_id == SYNTHETIC ? 2 * (s.n_cols / 8)
         : s.air_id == KECCAK_F ? KECCAK_N_CONSTRAINTS
         : s.air_id == KECCAK_SPONGE ? SPONGE_N_CONSTRAINTS
         : s.air_id == BYTE_PACKING ? PACK_N_CONSTRAINTS
         : s.air_id == MEMORY ? MEM_N_CONSTRAINTS
         : s.air_id == LOGIC ? LOGIC_N_CONSTRAINTS
         : s.air_id == PLONK ? PLONK_N_CONSTRAINTS
                                     : 2;
}
// the first aux column that is a running product (the columns before it are helpers); products run to the last column
// (AIR 8 keeps its two running products, columns 0 and 10, in its own kernels: stark_kernels.hip)
GL_HD uint32_t first_product(uint32_t air_id) { return air_id == KECCAK_F ? KECCAK_Z : air_id == MEMORY ? MEM_Z : 0; }

// sum_{j < n} beta^j col(j) by Horner from the top
template <class T, class Col>
GL_HD T compress(const Col& col, uint32_t n, T beta) {
  typedef Ops<T> F;
  T acc = col(n - 1);
#pragma unroll 1
  for (uint32_t j = n - 1; j-- > 0;) acc = F::add(F::mul(acc, beta), col(j));
  return acc;
}
// term of product column `col` (an aux column index) of the table: 1 + f (gamma + v - 1)
template <class T, class Row>
GL_HD T product_term(const Shape& s, uint32_t col, const uint64_t ctl[4], const Row& row) {
  typedef Ops<T> F;
  if (s.air_id == PLONK) return row.aux(col);  // the row's total num / den, left in the column by plonk_chunk_ratios_kernel
  if (s.air_id == SYNTHETIC) {
    const T beta = F::k(ctl[2 * (col & 1)]), gamma = F::k(ctl[2 * (col & 1) + 1]);
    return F::add(F::add(gamma, row.loc(8 * col)), F::mul(beta, row.loc(8 * col + 1)));
  }
  if (s.air_id == KECCAK_F) {
    const uint32_t c = col - KECCAK_Z;
    const T beta = F::k(ctl[2 * c]), gamma = F::k(ctl[2 * c + 1]), b50 = F::k(gl::pow(ctl[2 * c], TUPLE_LIMBS));
    // output limb j of the row: the iota output for lane 0, else the chi output (what constraint family F9 hands on)
    const T out = compress<T>([&](uint32_t j) { return j < 2 ? row.loc(keccak::COL_APPP + j) : row.loc(keccak::COL_APP + j); },
                              TUPLE_LIMBS, beta);
    const T v = F::add(row.aux(KECCAK_H + c), F::mul(b50, out));
    return F::add(F::k(1), F::mul(row.loc(keccak::COL_G), F::sub(F::add(gamma, v), F::k(1))));
  }
  if (s.air_id == KECCAK_SPONGE && col >= SPONGE_LZ) {
    // keccak_sponge -> logic: the XOR of limbs 8m .. 8m + 7 of the rate with the block is one operation of the logic
    // table: (0, 0, 1 | rate-before limbs | block limbs | xored limbs), limbs past the 34th zero
    const uint32_t m = (col - SPONGE_LZ) >> 1, c = (col - SPONGE_LZ) & 1;
    const T beta = F::k(ctl[2 * c]), gamma = F::k(ctl[2 * c + 1]);
    T acc = F::k(0);  // Horner from the top: result limbs, input-1 limbs, input-0 limbs, then the flags (0, 0, 1)
#pragma unroll 1
    for (uint32_t j = 8; j-- > 0;) {
      const uint32_t l = 8 * m + j;
      acc = F::mul(acc, beta);
      if (l < 34) acc = F::add(acc, row.loc(keccak_sponge::COL_XORED + l));
    }
#pragma unroll 1
    for (uint32_t part = 0; part < 2; part++) {  // part 0: the block (input 1), part 1: the rate before (input 0)
      const uint32_t bits0 = part == 0 ? keccak_sponge::COL_BLOCK : keccak_sponge::COL_RATE;
#pragma unroll 1
      for (uint32_t j = 8; j-- > 0;) {
        const uint32_t l = 8 * m + j;
        T limb = F::k(0);
        if (l < 34) {
#pragma unroll 1
          for (uint32_t z = 32; z-- > 0;) limb = F::add(F::dbl(limb), row.loc(bits0 + 32 * l + z));
        }
        acc = F::add(F::mul(acc, beta), limb);
      }
    }
    acc = F::add(F::mul(acc, beta), F::k(1));  // is_xor
    acc = F::mul(F::mul(acc, beta), beta);      // is_or = is_and = 0
    const T f = F::add(row.loc(keccak_sponge::COL_FULL), row.loc(keccak_sponge::COL_FINAL));
    return F::add(F::k(1), F::mul(f, F::sub(F::add(gamma, acc), F::k(1))));
  }
  if (s.air_id == LOGIC) {
    const uint32_t c = col - LOGIC_Z;
    const T beta = F::k(ctl[2 * c]), gamma = F::k(ctl[2 * c + 1]);
    T acc = F::k(0);
#pragma unroll 1
    for (uint32_t j = 8; j-- > 0;) acc = F::add(F::mul(acc, beta), row.loc(logic::COL_RES + j));
#pragma unroll 1
    for (uint32_t part = 0; part < 2; part++) {
      const uint32_t bits0 = part == 0 ? logic::COL_IN1 : logic::COL_IN0;
#pragma unroll 1
      for (uint32_t j = 8; j-- > 0;) {
        T limb = F::k(0);
#pragma unroll 1
        for (uint32_t z = 32; z-- > 0;) limb = F::add(F::dbl(limb), row.loc(bits0 + 32 * j + z));
        acc = F::add(F::mul(acc, beta), limb);
      }
    }
#pragma unroll 1
    for (uint32_t j = 3; j-- > 0;) acc = F::add(F::mul(acc, beta), row.loc(logic::COL_OP + j));
    return F::add(F::k(1), F::mul(row.loc(logic::COL_G), F::sub(F::add(gamma, acc), F::k(1))));
  }
  if (s.air_id == KECCAK_SPONGE) {
    const uint32_t c = col - SPONGE_Z;
    const T beta = F::k(ctl[2 * c]), gamma = F::k(ctl[2 * c + 1]);
    // tuple order: xored rate (34 limbs), capacity (16), updated state (50)
    const T v = compress<T>([&](uint32_t j) {
      return j < 34 ? row.loc(keccak_sponge::COL_XORED + j)
             : j < 50 ? row.loc(keccak_sponge::COL_CAP + j - 34)
                      : row.loc(keccak_sponge::COL_UPDATED + j - 50);
    }, 2 * TUPLE_LIMBS, beta);
    const T f = F::add(row.loc(keccak_sponge::COL_FULL), row.loc(keccak_sponge::COL_FINAL));
    return F::add(F::k(1), F::mul(f, F::sub(F::add(gamma, v), F::k(1))));
  }
  if (s.air_id == BYTE_PACKING) {
    const uint32_t c = col - PACK_Z;
    const T beta = F::k(ctl[2 * c]), gamma = F::k(ctl[2 * c + 1]);
    const T v = compress<T>([&](uint32_t j) {
      return j == 0 ? row.loc(byte_packing::COL_READ)
             : j == 1 ? row.loc(byte_packing::COL_ADDR)
             : j == 2 ? row.loc(byte_packing::COL_TS)
                      : row.loc(byte_packing::COL_VAL + j - 3);
    }, WORD_TUPLE, beta);
    T f = F::k(0);  // the row has a length: it moves a word
#pragma unroll 1
    for (uint32_t j = 0; j < 32; j++) f = F::add(f, row.loc(byte_packing::COL_LEN + j));
    return F::add(F::k(1), F::mul(f, F::sub(F::add(gamma, v), F::k(1))));
  }
  if (s.air_id == MEMORY) {
    const uint32_t c = col - MEM_Z;
    const T beta = F::k(ctl[2 * c]), gamma = F::k(ctl[2 * c + 1]);
    const T v = compress<T>([&](uint32_t j) {
      return j == 0 ? row.loc(memory::COL_READ)
             : j == 1 ? row.loc(memory::COL_ADDR)
             : j == 2 ? row.loc(memory::COL_TS)
                      : row.loc(memory::COL_VAL + j - 3);
    }, WORD_TUPLE, beta);
    return F::add(F::k(1), F::mul(row.loc(memory::COL_G), F::sub(F::add(gamma, v), F::k(1))));
  }
  return F::k(1);
}
// What ctl::eval keeps of a product's term when the table's AIR units emit the limb shares of the tuple themselves
// (keccak_sponge -> logic, both sides: logic::eval_unit, keccak_sponge::eval_unit): the flags' part.  Every other
// product: the whole term.
template <class T, class Row>
GL_HD T product_term_in_eval(const Shape& s, uint32_t col, const uint64_t ctl[4], const Row& row) {
  typedef Ops<T> F;
  if (s.air_id == KECCAK_SPONGE && col >= SPONGE_LZ) {
    const uint32_t c = (col - SPONGE_LZ) & 1;
    const T b2 = F::k(gl::mulc(ctl[2 * c], ctl[2 * c])), gamma = F::k(ctl[2 * c + 1]);   // is_xor = 1 sits at position 2
    const T f = F::add(row.loc(keccak_sponge::COL_FULL), row.loc(keccak_sponge::COL_FINAL));
    return F::add(F::k(1), F::mul(f, F::sub(F::add(gamma, b2), F::k(1))));
  }
  if (s.air_id == LOGIC) {
    const uint32_t c = col - LOGIC_Z;
    const T beta = F::k(ctl[2 * c]), gamma = F::k(ctl[2 * c + 1]);
    T acc = F::k(0);
#pragma unroll 1
    for (uint32_t j = 3; j-- > 0;) acc = F::add(F::mul(acc, beta), row.loc(logic::COL_OP + j));
    return F::add(F::k(1), F::mul(row.loc(logic::COL_G), F::sub(F::add(gamma, acc), F::k(1))));
  }
  return product_term<T>(s, col, ctl, row);
}
// AIR 8: the copy constraints of the PLONK-shaped circuit (plonky2's permutation argument: Z and partial products;
// upstream's vanishing-polynomial terms check_partial_products + L_1 (Z - 1)).  With the routed wires w_j, j < 80, in
// chunks of eight, per challenge set (beta, gamma):
//   num_k = prod_{j in chunk k} (w_j + beta k_j x + gamma)      k_j = 7^j, x = the row's point
//   den_k = prod_{j in chunk k} (w_j + beta sigma_j + gamma)
// and the running product goes BACKWARDS through the rows (as every running product of this library does): with
// cur_0 = Z(next row), cur_k = the k-th partial product (k = 1..9), cur_10 = Z(this row):
//   index base + 11 c + k - 1   all rows (the wrap from the last row to the first included)   cur_k den_k - cur_(k-1) num_k   deg 9
//   index base + 11 c + 10      first row                                                      Z - 1
// Z(first) = 1 and the cyclic relation force prod_rows prod_j (numerator / denominator) = 1.
// Challenge set c is the unit [10 c, 10 c + 10) of the auxiliary columns.
// The products are taken in groups of four independent multiplications (Ops::mul4 / mul4_lazy: only the last level
// needs canonical results): beta sigma_j for four wires (beta x k_j is a chain of multiplications by 7), then the pair products of numerator and denominator terms, then the halves; the chunk's last two levels
// last level (cur den, prev num) rides in the next chunk's last group, so every group is full.
// (Evaluated by plonk::eval_unit, chunk by chunk next to the gates that read the same wires.)

// The lookup part of the constraint list.  Synthetic tables: products [k0, k1) (their unit slicing); every other
// table: everything (one unit).
template <class T, class Row, class Emit>
GL_HD void eval(const Shape& s, uint32_t base, uint32_t k0, uint32_t k1, const uint64_t ctl[4], const Row& row, Emit& out) {
  typedef Ops<T> F;
  if (s.air_id == PLONK) return;  // the copy constraints ride in the AIR's own units (plonk::eval_unit)
  if (s.air_id == SYNTHETIC) {
#pragma unroll 1
    for (uint32_t k = k0; k < k1; k++) {
      const T z = row.aux(k), zn = row.aux_nxt(k), term = product_term<T>(s, k, ctl, row);
      out.transition(base + 2 * k, F::sub(z, F::mul(zn, term)));
      out.last(base + 2 * k + 1, F::sub(z, term));
    }
    return;
  }
  // Real tables: the unit that starts at column 0 also holds what is not a product (filter bits, the Keccak-f table's
  // carried input); the product columns [k0, k1) are this unit's (the sponge table's twelve products are sliced over
  // several units: each of the ten into the logic table sums 512 bit columns).
  uint32_t idx = base;
  const bool head = k0 == 0;
  if (s.air_id == KECCAK_F) {
    if (head) {
      const T g = row.loc(keccak::COL_G), s0 = row.loc(keccak::COL_STEP), s23 = row.loc(keccak::COL_STEP + 23);
      out.all(idx, F::sub(F::mul(g, g), g));
      out.all(idx + 1, F::mul(g, F::sub(F::k(1), s23)));
#pragma unroll 1
      for (uint32_t c = 0; c < 2; c++) {
        const T beta = F::k(ctl[2 * c]), h = row.aux(KECCAK_H + c);
        const T in = compress<T>([&](uint32_t j) { return row.loc(keccak::COL_A + j); }, TUPLE_LIMBS, beta);
        out.all(idx + 2 + 2 * c, F::mul(s0, F::sub(h, in)));
        out.transition(idx + 3 + 2 * c, F::mul(F::sub(F::k(1), s23), F::sub(row.aux_nxt(KECCAK_H + c), h)));
      }
    }
    idx += 6;
  }
  if (s.air_id == MEMORY || s.air_id == LOGIC) {
    if (head) {
      const T g = row.loc(s.air_id == MEMORY ? memory::COL_G : logic::COL_G);
      out.all(idx, F::sub(F::mul(g, g), g));
    }
    idx += 1;
  }
  const uint32_t p0 = first_product(s.air_id), p1 = n_aux(s);
#pragma unroll 1
  for (uint32_t k = k0 > p0 ? k0 : p0; k < (k1 < p1 ? k1 : p1); k++) {
    const T z = row.aux(k), zn = row.aux_nxt(k), term = product_term_in_eval<T>(s, k, ctl, row);
    out.transition(idx + 2 * (k - p0), F::sub(z, F::mul(zn, term)));
    out.last(idx + 2 * (k - p0) + 1, F::sub(z, term));
  }
}

// The lookups between tables: the product of the first-row values of the looking columns equals that of the looked
// columns, for challenge set c.  Tables by their position in a transaction (prover_state.rs:85-93).
struct Pair {
  const char* name;
  // the looking side: n_looking aux columns of challenge set 0, `stride` apart (set c is column + c): their first-row
  // values are multiplied together
  uint32_t looking_table, looking_air, looking_col, n_looking, stride;
  uint32_t looked_table, looked_air, looked_col;
};
constexpr uint32_t N_PAIRS = 3;
inline const Pair* pairs() {
  static const Pair P[N_PAIRS] = {
      {"keccak_sponge -> keccak_f", 4, KECCAK_SPONGE, SPONGE_Z, 1, 2, 3, KECCAK_F, KECCAK_Z},
      {"byte_packing -> memory", 1, BYTE_PACKING, PACK_Z, 1, 2, 6, MEMORY, MEM_Z},
      {"keccak_sponge -> logic", 4, KECCAK_SPONGE, SPONGE_LZ, SPONGE_LOGIC_OPS, 2, 5, LOGIC, LOGIC_Z},
  };
  return P;
}
}  // namespace ctl

struct Info {
  uint32_t air_id;
  const char* name;
  uint32_t n_cols;       // 0: any width (multiple of 4 not required; groups of four columns are constrained)
  uint32_t n_const_max;  // preprocessed constant columns the AIR can use
  uint32_t degree;       // constraint degree (x deg_pow for the synthetic AIR); rate_bits must give 2^r >= degree - 1
};
inline const Info* info(uint32_t air_id) {
  static const Info table[COUNT] = {
      {SYNTHETIC, "synthetic", 0, 4096, 3},
      {KECCAK_F, "keccak_f", keccak::N_COLS, 0, 3},
      {LOGIC, "logic", logic::N_COLS, 0, 3},
      {MEMORY, "memory", memory::N_COLS, 0, 3},
      {ARITHMETIC, "arithmetic", arithmetic::N_COLS, 0, 2},
      {BYTE_PACKING, "byte_packing", byte_packing::N_COLS, 0, 2},
      {KECCAK_SPONGE, "keccak_sponge", keccak_sponge::N_COLS, 0, 2},
      {ARITHMETIC_MUL, "arithmetic_mul", arithmetic_mul::N_COLS, 0, 3},
      {PLONK, "plonk", plonk::N_COLS, plonk::N_CONST, 9},
  };
  return air_id < COUNT ? &table[air_id] : nullptr;
}

}  // namespace air
}  // namespace bpg
