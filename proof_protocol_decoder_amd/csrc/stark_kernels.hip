// stark_kernels.hip -- K5/K6/K8/K9 and the glue kernels of the per-table STARK prover.
//
// Device side of prove_single_table / PolynomialBatch::prove_openings / fri_proof (upstream
// plonky2_evm / plonky2 @ 265d46a9, reached from plonky_block_proof_gen/src/proof_gen.rs:44-52).
// The AIR is the synthetic one of DESIGN.md section 4.  Layouts: include/bpg.h.
// MI355X-first choices that differ from upstream's CPU code but give identical field values:
//   * quotient: one lane per LDE row, constraints split into column chunks (Horner partials
//     recombined with alpha powers) so short-but-wide tables still fill 256 CUs;
//   * openings: dot product of each bit-reversed coefficient column with a power vector
//     zeta^bitrev(pos) (no Horner dependency chain, coalesced);
//   * FRI: alpha-combination in coefficient space (one pass over all coefficients), division by
//     (X - z) pointwise on the coset, folding in the EVALUATION domain (upstream folds
//     coefficients and re-FFTs every layer).
#include "common.hpp"
#include "gl.hpp"
#include "poseidon.cuh"
#include "poseidon_mx.cuh"
#include "stark_kernels.hpp"

namespace {

using gl::Ext;

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t rnd(uint64_t seed, uint64_t col, uint64_t row) {
  return gl::canon(splitmix64(seed ^ (col << 32) ^ row));
}
__device__ __forceinline__ uint64_t pow_e(uint64_t t, uint32_t e) {
  return e == 3 ? gl::mulc(gl::mulc(t, t), t) : t;
}
// w_n^m from the half-size table tw[e] = w_n^e, e < n/2
__device__ __forceinline__ uint64_t root_pow(const uint64_t* __restrict__ tw, uint32_t log_n, uint32_t m) {
  if (log_n == 0) return 1;
  const uint32_t half = 1u << (log_n - 1);
  uint64_t w = tw[m & (half - 1)];
  return (m & half) ? gl::negc(w) : w;
}

// ---------------------------------------------------------------- synthetic witness
__global__ void __launch_bounds__(256)
synth_const_kernel(uint64_t* out, uint32_t log_n, uint32_t n_const, uint64_t seed) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= ((uint64_t)n_const << log_n)) return;
  out[i] = rnd(seed ^ 0xC0115700C0115700ULL, i >> log_n, i & ((1ull << log_n) - 1));
}
// grid = (rows/256, groups + 1, proofs): group g < G fills columns 4g..4g+3; blockIdx.y == G fills the tail.
__global__ void __launch_bounds__(256)
synth_trace_kernel(bpg::BatchOf<bpg::SynthTraceArgs> batch, uint32_t log_n, uint32_t n_cols, uint32_t n_const,
                   uint32_t deg_pow) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  uint64_t* __restrict__ t = batch.a[blockIdx.z].trace;
  const uint64_t* __restrict__ consts = batch.a[blockIdx.z].consts;
  const uint64_t seed = batch.a[blockIdx.z].seed;
  const uint32_t n = 1u << log_n, G = n_cols / 4;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t g = blockIdx.y;
  if (g == G) {
    for (uint32_t c = 4 * G; c < n_cols; c++) t[(uint64_t)c * n + i] = rnd(seed, c, i);
    return;
  }
  uint64_t* a = t + (uint64_t)(4 * g) * n;
  const uint64_t av = rnd(seed, 4 * g, i), bv = rnd(seed, 4 * g + 1, i);
  const uint64_t q = n_const ? consts[(uint64_t)(g % n_const) * n + i] : 1;
  const uint64_t ab = gl::mulc(av, bv);
  const uint64_t cv = gl::addc(ab, gl::mulc(q, av));
  a[i] = av;
  a[n + i] = bv;
  a[2 * (uint64_t)n + i] = cv;
  if (i == 0) a[3 * (uint64_t)n] = gl::addc(av, bv);
  if (i + 1 < n) a[3 * (uint64_t)n + i + 1] = gl::addc(pow_e(gl::mulc(ab, cv), deg_pow), bv);
}

// ---------------------------------------------------------------- Keccak-f witness (AIR 1, air.hpp)
// Row r = round r % 24 of permutation r / 24.  A lane owns one row: it replays the earlier rounds of its
// permutation in registers (at most 23 rounds of 64-bit logic, nothing next to the 2430 words it then stores) so
// that every store of the launch is coalesced across rows.  grid.y splits the columns: 0 = flags, limbs, C, C',
// A''; 1..5 = the A' bits of sheet row y - 1.  Input lanes: `inputs` ([permutation][25], any u64) or, when null,
// splitmix64(seed ^ (lane << 32) ^ permutation) -- the same stream the oracle draws.
__global__ void __launch_bounds__(256)
keccak_trace_kernel(uint64_t* __restrict__ t, const uint64_t* __restrict__ inputs, uint32_t log_n, uint64_t seed) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  namespace kk = bpg::air::keccak;
  const uint32_t n = 1u << log_n;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t perm = i / 24, rnd_no = i % 24;
  uint64_t a[25];
  for (uint32_t l = 0; l < 25; l++)
    a[l] = inputs ? inputs[(uint64_t)perm * 25 + l] : splitmix64(seed ^ ((uint64_t)l << 32) ^ perm);
  kk::Round R;
  for (uint32_t r = 0; r < rnd_no; r++) {
    kk::round(a, r, R);
    for (uint32_t l = 0; l < 25; l++) a[l] = R.app[l];
    a[0] = R.appp0;
  }
  kk::round(a, rnd_no, R);
  auto put = [&](uint32_t col, uint64_t v) { t[(uint64_t)col * n + i] = v; };
  if (blockIdx.y == 0) {
    for (uint32_t k = 0; k < 24; k++) put(kk::COL_STEP + k, k == rnd_no);
    for (uint32_t l = 0; l < 25; l++) {
      put(kk::COL_A + 2 * l, (uint32_t)a[l]);
      put(kk::COL_A + 2 * l + 1, a[l] >> 32);
      put(kk::COL_APP + 2 * l, (uint32_t)R.app[l]);
      put(kk::COL_APP + 2 * l + 1, R.app[l] >> 32);
    }
    for (uint32_t x = 0; x < 5; x++)
      for (uint32_t z = 0; z < 64; z++) {
        put(kk::COL_C + 64 * x + z, (R.c[x] >> z) & 1);
        put(kk::COL_CP + 64 * x + z, (R.cp[x] >> z) & 1);
      }
    for (uint32_t z = 0; z < 64; z++) put(kk::COL_APP0_BITS + z, (R.app[0] >> z) & 1);
    put(kk::COL_APPP, (uint32_t)R.appp0);
    put(kk::COL_APPP + 1, R.appp0 >> 32);
    put(kk::COL_G, 0);  // nothing exposed to the lookup until lookup_filter_kernel says otherwise
  } else {
    const uint32_t y = blockIdx.y - 1;
    for (uint32_t x = 0; x < 5; x++) {
      const uint64_t v = R.ap[x + 5 * y];
      for (uint32_t z = 0; z < 64; z++) put(kk::COL_AP + 64 * (x + 5 * y) + z, (v >> z) & 1);
    }
  }
}

// ---------------------------------------------------------------- logic witness (AIR 2, air.hpp)
// One operation per row.  `inputs` ([row][9]: operation code 0 none / 1 and / 2 or / 3 xor, then the four 64-bit words
// of each operand, least significant first) or, when null, drawn from the seed -- the same stream the oracle draws:
// code = splitmix64(seed ^ (0xFF << 32) ^ row) & 3, word w of operand j = splitmix64(seed ^ ((1 + 4 j + w) << 32) ^ row).
// grid.y = 0: flags, result limbs and the bits of operand 0; 1: the bits of operand 1.  Stores coalesce across rows.
__global__ void __launch_bounds__(256)
logic_trace_kernel(uint64_t* __restrict__ t, const uint64_t* __restrict__ inputs, uint32_t log_n, uint64_t seed) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  namespace lg = bpg::air::logic;
  const uint32_t n = 1u << log_n;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t op = (uint32_t)(inputs ? inputs[(uint64_t)i * 9] : splitmix64(seed ^ (0xFFull << 32) ^ i)) & 3u;
  uint64_t w[2][4];
  for (uint32_t j = 0; j < 2; j++)
    for (uint32_t k = 0; k < 4; k++)
      w[j][k] = inputs ? inputs[(uint64_t)i * 9 + 1 + 4 * j + k] : splitmix64(seed ^ ((uint64_t)(1 + 4 * j + k) << 32) ^ i);
  auto put = [&](uint32_t col, uint64_t v) { t[(uint64_t)col * n + i] = v; };
  const uint32_t j = blockIdx.y;
  if (j == 0) {
    put(lg::COL_OP, op == lg::OP_AND);
    put(lg::COL_OP + 1, op == lg::OP_OR);
    put(lg::COL_OP + 2, op == lg::OP_XOR);
    for (uint32_t k = 0; k < 8; k++) {
      const uint32_t a = (uint32_t)(w[0][k >> 1] >> (32 * (k & 1))), b = (uint32_t)(w[1][k >> 1] >> (32 * (k & 1)));
      put(lg::COL_RES + k, lg::apply(op, a, b));
    }
    put(lg::COL_G, 0);  // the lookup's filter: set by logic_lookup_filter_kernel where the sponge table asks
  }
  for (uint32_t z = 0; z < 256; z++) put((j ? lg::COL_IN1 : lg::COL_IN0) + z, (w[j][z >> 6] >> (z & 63)) & 1);
}

// ---------------------------------------------------------------- memory witness (AIR 3, air.hpp)
// One operation per row, rows sorted by (address, timestamp).  `inputs` ([row][11]: is_read, address, timestamp, eight
// value limbs; already sorted -- an unsorted log yields a witness the verifier rejects) or, when null, a log drawn
// from the seed, four operations per address, the same the oracle draws (h(c, i) = splitmix64(seed ^ (c << 32) ^ i)):
//   address of group g = 4 g + (h(0xA0, g) & 3), timestamp of row i = 8 i + (h(0xA1, i) & 7), is_read = h(0xA2, i) & 1,
//   a write stores limbs h(0xB0 + k, i) & 0xFFFFFFFF, a read returns what the group's previous operation left (0 first).
__global__ void __launch_bounds__(256)
memory_trace_kernel(uint64_t* __restrict__ t, const uint64_t* __restrict__ inputs, uint32_t log_n, uint64_t seed) {
  if (gridDim.x <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  namespace mm = bpg::air::memory;
  const uint32_t n = 1u << log_n;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  auto h = [&](uint64_t c, uint64_t x) { return splitmix64(seed ^ (c << 32) ^ x); };
  uint64_t rd, addr, ts, v[8], addr_n = 0, ts_n = 0;
  if (inputs) {
    const uint64_t* r = inputs + (uint64_t)i * 11;
    rd = r[0] & 1; addr = r[1]; ts = r[2];
    for (uint32_t k = 0; k < 8; k++) v[k] = r[3 + k];
    if (i + 1 < n) { addr_n = r[11 + 1]; ts_n = r[11 + 2]; }
  } else {
    const uint32_t g = i >> 2, j = i & 3;
    addr = 4ull * g + (h(0xA0, g) & 3);
    addr_n = j == 3 ? 4ull * (g + 1) + (h(0xA0, g + 1) & 3) : addr;
    ts = 8ull * i + (h(0xA1, i) & 7);
    ts_n = 8ull * (i + 1) + (h(0xA1, i + 1) & 7);
    for (uint32_t k = 0; k < 8; k++) v[k] = 0;
    rd = 0;
    for (uint32_t jj = 0; jj <= j; jj++) {  // replay the group up to this row
      const uint32_t ii = 4 * g + jj;
      rd = h(0xA2, ii) & 1;
      if (!rd)
        for (uint32_t k = 0; k < 8; k++) v[k] = h(0xB0 + k, ii) & 0xFFFFFFFFull;
    }
  }
  const bool last = i + 1 == n, chg = !last && addr_n != addr;
  const uint32_t gap = last ? 0u : (uint32_t)(chg ? addr_n - addr - 1 : ts_n - ts - 1);
  auto put = [&](uint32_t col, uint64_t x) { t[(uint64_t)col * n + i] = x; };
  put(mm::COL_READ, rd);
  put(mm::COL_ADDR, gl::canon(addr));
  put(mm::COL_TS, gl::canon(ts));
  for (uint32_t k = 0; k < 8; k++) put(mm::COL_VAL + k, gl::canon(v[k]));
  put(mm::COL_CHG, chg);
  for (uint32_t z = 0; z < 32; z++) put(mm::COL_GAP + z, (gap >> z) & 1);
  put(mm::COL_G, 0);  // nothing exposed to the lookup until launch_lookup_filter says otherwise
}

// ---------------------------------------------------------------- arithmetic witness (AIR 4, air.hpp)
// One operation per row.  `inputs` ([row][9]: operation code 0 none / 1 add / 2 sub / 3 lt / 4 gt (anything else:
// none), then the four 64-bit words of x and of y, least significant first) or, when null, drawn from the seed like the
// oracle: code = splitmix64(seed ^ (0xFE << 32) ^ row) % 5, word w of operand j = splitmix64(seed ^ ((1 + 4 j + w) << 32) ^ row).
__global__ void __launch_bounds__(256)
arithmetic_trace_kernel(uint64_t* __restrict__ t, const uint64_t* __restrict__ inputs, uint32_t log_n, uint64_t seed) {
  if (gridDim.x <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  namespace ar = bpg::air::arithmetic;
  const uint32_t n = 1u << log_n;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t code = inputs ? inputs[(uint64_t)i * 9] : splitmix64(seed ^ (0xFEull << 32) ^ i) % 5;
  const uint32_t op = code <= 4 ? (uint32_t)code : 0u;
  uint64_t x[4], y[4];
  for (uint32_t k = 0; k < 4; k++) {
    x[k] = inputs ? inputs[(uint64_t)i * 9 + 1 + k] : splitmix64(seed ^ ((uint64_t)(1 + k) << 32) ^ i);
    y[k] = inputs ? inputs[(uint64_t)i * 9 + 5 + k] : splitmix64(seed ^ ((uint64_t)(5 + k) << 32) ^ i);
  }
  auto limb = [](const uint64_t (&w)[4], uint32_t k) { return (uint32_t)(w[k >> 2] >> (16 * (k & 3))) & 0xFFFFu; };
  auto put = [&](uint32_t col, uint64_t v) { t[(uint64_t)col * n + i] = v; };
  put(ar::COL_OP, op == ar::OP_ADD);
  put(ar::COL_OP + 1, op == ar::OP_SUB);
  put(ar::COL_OP + 2, op == ar::OP_LT);
  put(ar::COL_OP + 3, op == ar::OP_GT);
  // the chain U + V = W + 2^256 c: add: z = x + y; sub / lt: z = x - y (z + y = x); gt: z = y - x (z + x = y)
  uint32_t carry = 0;
  for (uint32_t k = 0; k < 16; k++) {
    const uint32_t xk = limb(x, k), yk = limb(y, k);
    uint32_t zk, ck;
    if (op == ar::OP_ADD) {
      const uint32_t s = xk + yk + carry;
      zk = s & 0xFFFFu; ck = s >> 16;
    } else if (op == ar::OP_NONE) {
      zk = 0; ck = 0;
    } else {  // z = w - v with borrow; the chain's carry out of limb k is that borrow
      const uint32_t w = op == ar::OP_GT ? yk : xk, v = op == ar::OP_GT ? xk : yk;
      const uint32_t d = w - v - carry;
      zk = d & 0xFFFFu; ck = (d >> 16) & 1u;
    }
    put(ar::COL_X + k, xk);
    put(ar::COL_Y + k, yk);
    for (uint32_t j = 0; j < 16; j++) put(ar::COL_Z + 16 * k + j, (zk >> j) & 1);
    put(ar::COL_CARRY + k, ck);
    carry = ck;
  }
  put(ar::COL_RES, (op == ar::OP_LT || op == ar::OP_GT) ? carry : 0);
}

// ---------------------------------------------------------------- byte-packing witness (AIR 5, air.hpp)
// One sequence per row.  `inputs` ([row][6]: word 0 = is_read (bit 0) | timestamp << 8, word 1 = len (low byte: 0 = a
// padding row, else 1..32; larger values: 32) | address << 8 -- address and timestamp (32 bits each) of the memory
// operation that moves the word, zero for callers that do not care --, then the 32 byte slots as four 64-bit words,
// slot i = byte i % 8 of word i / 8; slots at or beyond len are ignored) or, when null, drawn from the seed like the
// oracle: is_read = h(0xC0) & 1, len = h(0xC1) % 33, word w = h(0xC2 + w), h(c) = splitmix64(seed ^ (c << 32) ^ row),
// address = row, timestamp = 2 + row.
__global__ void __launch_bounds__(256)
byte_packing_trace_kernel(uint64_t* __restrict__ t, const uint64_t* __restrict__ inputs, uint32_t log_n, uint64_t seed) {
  if (gridDim.x <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  namespace bp = bpg::air::byte_packing;
  const uint32_t n = 1u << log_n;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  auto h = [&](uint64_t c) { return splitmix64(seed ^ (c << 32) ^ i); };
  const uint64_t w0 = inputs ? inputs[(uint64_t)i * 6] : 0, w1 = inputs ? inputs[(uint64_t)i * 6 + 1] : 0;
  const uint64_t rd = (inputs ? w0 : h(0xC0)) & 1;
  uint64_t len = inputs ? (w1 & 0xFF) : h(0xC1) % 33;
  if (len > 32) len = 32;
  const uint64_t addr = inputs ? (w1 >> 8) & 0xFFFFFFFFull : i, ts = inputs ? (w0 >> 8) & 0xFFFFFFFFull : 2ull + i;
  uint64_t w[4];
  for (uint32_t k = 0; k < 4; k++) w[k] = inputs ? inputs[(uint64_t)i * 6 + 2 + k] : h(0xC2 + k);
  auto put = [&](uint32_t col, uint64_t v) { t[(uint64_t)col * n + i] = v; };
  put(bp::COL_READ, rd);
  for (uint32_t j = 1; j <= 32; j++) put(bp::COL_LEN + j - 1, len == j);
  uint32_t limb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t s = 0; s < 32; s++) {
    const uint32_t byte = s < len ? (uint32_t)(w[s >> 3] >> (8 * (s & 7))) & 0xFFu : 0u;
    for (uint32_t b = 0; b < 8; b++) put(bp::COL_BITS + 8 * s + b, (byte >> b) & 1);
    if (s < len) {
      const uint32_t p = (uint32_t)len - 1 - s;  // weight 256^p of the big-endian value
      limb[p >> 2] |= byte << (8 * (p & 3));
    }
  }
  for (uint32_t k = 0; k < 8; k++) put(bp::COL_VAL + k, limb[k]);
  put(bp::COL_ADDR, addr);
  put(bp::COL_TS, ts);
}
// The memory log that goes with a byte-packing table (lookup byte_packing -> memory, air::ctl), as input for
// memory_trace_kernel: packing row r is two operations on its address, the write that put the word there (timestamp 1)
// and the operation the row looks up -- (is_read, address, timestamp, word) --; the rows of the memory table beyond
// those re-read the last address.  Addresses must ascend with the packing rows (seeded tables: address = row).
__global__ void __launch_bounds__(256)
memory_inputs_from_byte_packing_kernel(const uint64_t* __restrict__ pack, uint32_t pack_log_n, uint64_t* __restrict__ log,
                                       uint32_t n_mem) {
  namespace bp = bpg::air::byte_packing;
  const uint32_t P = 1u << pack_log_n, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_mem) return;
  const uint32_t r = (i >> 1) < P ? (i >> 1) : P - 1;
  const bool pad = (i >> 1) >= P, second = pad || (i & 1);
  uint64_t* o = log + (uint64_t)i * 11;
  const uint64_t ts = pack[(uint64_t)bp::COL_TS * P + r];
  o[0] = pad ? 1 : (second ? pack[(uint64_t)bp::COL_READ * P + r] : 0);
  o[1] = pack[(uint64_t)bp::COL_ADDR * P + r];
  o[2] = pad ? ts + (i - 2 * P + 1) : (second ? ts : 1);
  for (uint32_t k = 0; k < 8; k++) o[3 + k] = pack[(uint64_t)(bp::COL_VAL + k) * P + r];
}

// ---------------------------------------------------------------- Keccak sponge witness (AIR 6, air.hpp)
// One absorbed block per row.  `inputs` ([row][44]: flags (1 = full block, 2 = final block, 0 = padding row), the number
// of message bytes in the block, the block as absorbed (17 words, padding included), the 25 lanes of the state before
// the block) or, when null, one single-block message per row drawn from the seed like the oracle:
// h(c) = splitmix64(seed ^ (c << 32) ^ row); every eighth row (h(0xD3) % 8 == 0) is a padding row, else len = h(0xD0) % 136,
// message word w = h(0xD1 + (w << 8)), state before = 0; rows from row_limit on are padding rows (seeded tables only).
// The updated state is the permutation of (xored rate, capacity).
__global__ void __launch_bounds__(256)
keccak_sponge_trace_kernel(uint64_t* __restrict__ t, const uint64_t* __restrict__ inputs, uint32_t log_n, uint64_t seed,
                           uint32_t row_limit) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  namespace sp = bpg::air::keccak_sponge;
  namespace kk = bpg::air::keccak;
  const uint32_t n = 1u << log_n;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t flags, len, blk[17], st[25];
  if (inputs) {
    const uint64_t* r = inputs + (uint64_t)i * 44;
    flags = r[0] & 3; len = r[1];
    for (uint32_t w = 0; w < 17; w++) blk[w] = r[2 + w];
    for (uint32_t l = 0; l < 25; l++) st[l] = r[19 + l];
  } else {
    auto h = [&](uint64_t c) { return splitmix64(seed ^ (c << 32) ^ i); };
    for (uint32_t l = 0; l < 25; l++) st[l] = 0;
    if (h(0xD3) % 8 == 0 || i >= row_limit) {  // row_limit: a seeded table asks for no more permutations than its Keccak-f table holds
      flags = 0; len = 0;
      for (uint32_t w = 0; w < 17; w++) blk[w] = 0;
    } else {
      flags = 2; len = h(0xD0) % 136;
      for (uint32_t w = 0; w < 17; w++) {
        const uint64_t m = h(0xD1 + ((uint64_t)w << 8));
        const uint32_t lo = 8 * w;  // bytes lo .. lo + 7 of the block: keep the first len - lo of them
        blk[w] = len >= lo + 8 ? m : len > lo ? m & ((1ull << (8 * (len - lo))) - 1) : 0;
      }
      blk[len >> 3] ^= 1ull << (8 * (len & 7));
      blk[16] ^= 0x80ull << 56;
    }
  }
  if (flags == 3) flags = 0;
  uint64_t a[25];
  for (uint32_t l = 0; l < 25; l++) a[l] = l < 17 ? st[l] ^ blk[l] : st[l];
  kk::Round R;
  uint64_t p[25];
  for (uint32_t l = 0; l < 25; l++) p[l] = a[l];
  if (flags) {
    for (uint32_t r = 0; r < 24; r++) {
      kk::round(p, r, R);
      for (uint32_t l = 0; l < 25; l++) p[l] = R.app[l];
      p[0] = R.appp0;
    }
  } else {
    for (uint32_t l = 0; l < 25; l++) p[l] = 0;
  }
  auto put = [&](uint32_t col, uint64_t v) { t[(uint64_t)col * n + i] = v; };
  const uint32_t part = blockIdx.y;  // 0: flags, lengths, limbs; 1: block bits; 2: rate bits
  if (part == 0) {
    put(sp::COL_FULL, flags == 1);
    put(sp::COL_FINAL, flags == 2);
    for (uint32_t j = 0; j < 136; j++) put(sp::COL_LEN + j, flags == 2 && len == j);
    for (uint32_t k = 0; k < 16; k++) put(sp::COL_CAP + k, (uint32_t)(st[17 + k / 2] >> (32 * (k & 1))));
    for (uint32_t k = 0; k < 34; k++) put(sp::COL_XORED + k, (uint32_t)(a[k / 2] >> (32 * (k & 1))));
    for (uint32_t k = 0; k < 50; k++) put(sp::COL_UPDATED + k, (uint32_t)(p[k / 2] >> (32 * (k & 1))));
  } else if (part == 1) {
    for (uint32_t z = 0; z < 1088; z++) put(sp::COL_BLOCK + z, (blk[z >> 6] >> (z & 63)) & 1);
  } else {
    for (uint32_t z = 0; z < 1088; z++) put(sp::COL_RATE + z, (st[z >> 6] >> (z & 63)) & 1);
  }
}

// The permutations a transaction's SEEDED Keccak-f table holds when its sponge table is real too: permutation p is the
// one sponge row p asks for -- input = that row's (xored rate, capacity) -- wherever row p absorbs a block; every other
// permutation is drawn from the seed as keccak_trace_kernel would.  inputs: [n_perms][25] lanes.  This is what makes the
// two seeded tables one statement for the lookup keccak_sponge -> keccak_f (air::ctl).
__global__ void __launch_bounds__(256)
keccak_inputs_from_sponge_kernel(const uint64_t* __restrict__ sponge, uint32_t sponge_log_n, uint64_t* __restrict__ inputs,
                                 uint32_t n_perms, uint64_t seed) {
  namespace sp = bpg::air::keccak_sponge;
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x, n = 1u << sponge_log_n;
  if (p >= n_perms) return;
  const bool asked = p < n && (sponge[(uint64_t)sp::COL_FULL * n + p] + sponge[(uint64_t)sp::COL_FINAL * n + p]) != 0;
  for (uint32_t l = 0; l < 25; l++) {
    uint64_t v;
    if (asked) {
      const uint32_t c = l < 17 ? sp::COL_XORED + 2 * l : sp::COL_CAP + 2 * (l - 17);
      v = sponge[(uint64_t)c * n + p] | (sponge[(uint64_t)(c + 1) * n + p] << 32);
    } else {
      v = splitmix64(seed ^ ((uint64_t)l << 32) ^ p);
    }
    inputs[(uint64_t)p * 25 + l] = v;
  }
}

// The operations a transaction's logic table holds when its sponge table is real too (keccak_sponge -> logic, air::ctl):
// rows 5 p + m, p < covered sponge rows, are the XOR of limbs 8 m .. 8 m + 7 of sponge row p's rate with its block where
// row p absorbs a block (a padding operation where it does not); the rows after them are the caller's operations
// (`given`, n_given of them, then padding) or, without any, drawn from the seed as logic_trace_kernel would draw them.
// inputs: [n_logic][9] = code, operand 0 (rate before), operand 1 (block), as logic_trace_kernel reads them.
__global__ void __launch_bounds__(256)
logic_inputs_from_sponge_kernel(const uint64_t* __restrict__ sponge, uint32_t sponge_log_n, uint32_t covered,
                                const uint64_t* __restrict__ given, uint32_t n_given, uint64_t* __restrict__ inputs, uint32_t n_logic,
                                uint64_t seed) {
  namespace sp = bpg::air::keccak_sponge;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n = 1u << sponge_log_n;
  if (i >= n_logic) return;
  uint64_t* o = inputs + (uint64_t)i * 9;
  const uint32_t region = 5 * covered;
  if (i < region) {
    const uint32_t p = i / 5, m = i % 5;
    const bool asked = (sponge[(uint64_t)sp::COL_FULL * n + p] + sponge[(uint64_t)sp::COL_FINAL * n + p]) != 0;
    o[0] = asked ? bpg::air::logic::OP_XOR : bpg::air::logic::OP_NONE;
    for (uint32_t j = 0; j < 2; j++)
      for (uint32_t w = 0; w < 4; w++) {
        uint64_t v = 0;
        if (asked)
          for (uint32_t z = 0; z < 64; z++) {
            const uint32_t bit = 256 * m + 64 * w + z;  // bit of the 1088-bit rate / block
            if (bit < 1088) v |= (sponge[(uint64_t)((j ? sp::COL_BLOCK : sp::COL_RATE) + bit) * n + p] & 1) << z;
          }
        o[1 + 4 * j + w] = v;
      }
    return;
  }
  const uint32_t k = i - region;
  if (given) {
    for (uint32_t w = 0; w < 9; w++) o[w] = k < n_given ? given[(uint64_t)k * 9 + w] : 0;
  } else {
    o[0] = splitmix64(seed ^ (0xFFull << 32) ^ i) & 3u;
    for (uint32_t w = 0; w < 8; w++) o[1 + w] = splitmix64(seed ^ ((uint64_t)(1 + w) << 32) ^ i);
  }
}
// ... and the filter of that table: g = 1 on the rows 5 p + m whose sponge row p absorbs a block
__global__ void __launch_bounds__(256)
logic_lookup_filter_kernel(uint64_t* __restrict__ trace, uint32_t log_n, const uint64_t* __restrict__ full, const uint64_t* __restrict__ fin,
                           uint32_t covered) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n = 1u << log_n;
  if (i >= 5 * covered || i >= n) return;
  trace[(uint64_t)bpg::air::logic::COL_G * n + i] = (full[i / 5] + fin[i / 5]) != 0;
}

// ---------------------------------------------------------------- multiplication witness (AIR 7, air.hpp)
// One product per row.  `inputs` ([row][9]: is_mul, the four 64-bit words of x and of y, least significant first) or,
// when null, drawn from the seed like the oracle: is_mul = splitmix64(seed ^ (0xE0 << 32) ^ row) % 4 != 0, words as in
// bp_logic_trace.  Schoolbook over 16-bit limbs: column sums stay below 2^37.
__global__ void __launch_bounds__(256)
arithmetic_mul_trace_kernel(uint64_t* __restrict__ t, const uint64_t* __restrict__ inputs, uint32_t log_n, uint64_t seed) {
  if (gridDim.x <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  namespace am = bpg::air::arithmetic_mul;
  const uint32_t n = 1u << log_n;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool mul = inputs ? (inputs[(uint64_t)i * 9] & 1) != 0 : splitmix64(seed ^ (0xE0ull << 32) ^ i) % 4 != 0;
  uint64_t x[4], y[4];
  for (uint32_t k = 0; k < 4; k++) {
    x[k] = inputs ? inputs[(uint64_t)i * 9 + 1 + k] : splitmix64(seed ^ ((uint64_t)(1 + k) << 32) ^ i);
    y[k] = inputs ? inputs[(uint64_t)i * 9 + 5 + k] : splitmix64(seed ^ ((uint64_t)(5 + k) << 32) ^ i);
  }
  auto limb = [](const uint64_t (&w)[4], uint32_t k) { return (uint64_t)((w[k >> 2] >> (16 * (k & 3))) & 0xFFFFu); };
  auto put = [&](uint32_t col, uint64_t v) { t[(uint64_t)col * n + i] = v; };
  put(am::COL_MUL, mul);
  for (uint32_t k = 0; k < 16; k++) {
    put(am::COL_X + k, limb(x, k));
    put(am::COL_Y + k, limb(y, k));
  }
  uint64_t carry = 0;
  for (uint32_t k = 0; k < 32; k++) {
    uint64_t sum = carry;
    if (mul) {
      const uint32_t i0 = k < 16 ? 0 : k - 15, i1 = k < 16 ? k : 15;
      for (uint32_t a = i0; a <= i1; a++) sum += limb(x, a) * limb(y, k - a);
    }
    const uint32_t p = (uint32_t)(sum & 0xFFFFu);
    carry = sum >> 16;
    const uint32_t pcol = k < 16 ? am::COL_Z + 16 * k : am::COL_W + 16 * (k - 16);
    for (uint32_t j = 0; j < 16; j++) put(pcol + j, (p >> j) & 1);
    for (uint32_t j = 0; j < 21; j++) put(am::COL_CARRY + 21 * k + j, (carry >> j) & 1);
  }
}

// ---------------------------------------------------------------- auxiliary columns: the table's lookups (air::ctl)
// A row of the trace as the lookup terms read it (values on the trace domain, column-major).
struct TraceRow {
  const uint64_t *trace, *aux_;
  uint64_t n, i;
  __device__ __forceinline__ uint64_t x() const { return 0; }  // (no product term reads the point or the constants)
  __device__ __forceinline__ uint64_t cst(uint32_t) const { return 0; }
  __device__ __forceinline__ uint64_t loc(uint32_t c) const { return trace[(uint64_t)c * n + i]; }
  __device__ __forceinline__ uint64_t aux(uint32_t k) const { return aux_[(uint64_t)k * n + i]; }
};
// Running products  z_k[i] = prod_{i' >= i} term_k[i'],  term = air::ctl::product_term (synthetic tables: gamma + a +
// beta * b over trace columns 8k, 8k + 1; real tables: 1 + filter * (gamma + compressed tuple - 1)).
// One workgroup per product column.  The column is walked back to front in tiles of 8*T elements; inside a
// tile lane t owns the 8 consecutive elements [8t, 8t+8) (one 64-byte piece: a wave covers 4 KiB of
// each input per tile, read and written once with 16-byte accesses), the per-lane products are
// suffix-scanned through LDS and multiplied by the product of the tiles already done.
// (The first version gave each lane one n/T-element chunk: lanes n/T*8 bytes apart, every access a
// different cache line, 16x the algorithmic HBM reads on the 2^14..2^17-row tables by PMC.)
template <uint32_t AIR>
__global__ void __launch_bounds__(1024)
aux_suffix_product_kernel(bpg::BatchOf<bpg::AuxArgs> batch, uint32_t log_n, uint32_t n_cols) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  const uint64_t* __restrict__ trace = batch.a[blockIdx.z].trace;
  uint64_t* __restrict__ aux = batch.a[blockIdx.z].aux;
  const bpg::Ctl& ctl = batch.a[blockIdx.z].ctl;
  __shared__ uint64_t part[1024];
  const bpg::air::Shape shape{AIR, n_cols, 0, 1};
  const uint32_t n = 1u << log_n, T = blockDim.x, t = threadIdx.x;
  const uint32_t k = AIR == bpg::air::PLONK ? 10 * blockIdx.x : bpg::air::ctl::first_product(AIR) + blockIdx.x;  // (plonk: Z_0, Z_1)
  uint64_t* z = aux + (uint64_t)k * n;
  // elements per lane per tile (n and T are powers of two; the launcher never uses fewer than 64
  // lanes, so T may exceed n: the surplus lanes then hold the neutral element)
  const uint32_t per0 = n >= 8 * T ? 8 : (n >= T ? n / T : 1);
  const uint32_t tile = n >= T ? per0 * T : n;
  const uint32_t per = t * per0 < tile ? per0 : 0;       // 0: this lane has no elements
  uint64_t done = 1;                                     // product of every tile after the current one
  for (uint32_t base = n; base > 0;) {
    base -= tile;
    const uint32_t lo = base + t * per0;
    uint64_t f[8], p = 1;
#pragma unroll
    for (int j = 7; j >= 0; j--) {
      if ((uint32_t)j < per) {
        f[j] = bpg::air::ctl::product_term<uint64_t>(shape, k, ctl.v, TraceRow{trace, aux, n, lo + (uint32_t)j});
        p = gl::mulc(p, f[j]);
      }
    }
    part[t] = p;
    __syncthreads();
    // inclusive suffix scan over the per-lane products (Hillis-Steele)
    for (uint32_t d = 1; d < T; d <<= 1) {
      const uint64_t v = part[t];
      const uint64_t o = t + d < T ? part[t + d] : 1;
      __syncthreads();
      part[t] = gl::mulc(v, o);
      __syncthreads();
    }
    uint64_t carry = gl::mulc(t + 1 < T ? part[t + 1] : 1, done);  // everything after my piece
    const uint64_t whole = part[0];
    __syncthreads();  // part is rewritten by the next tile
#pragma unroll
    for (int j = 7; j >= 0; j--) {
      if ((uint32_t)j < per) {
        carry = gl::mulc(carry, f[j]);
        f[j] = carry;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; j++)
      if ((uint32_t)j < per) z[lo + j] = f[j];
    done = gl::mulc(done, whole);
  }
}
// The helper columns of the Keccak-f table's lookup (air::ctl): h_0 / h_1, the permutation's input compressed by beta_0 /
// beta_1 and carried along its rows.  A lane owns a row; it reads the 50 input limbs of its permutation's first row.
__global__ void __launch_bounds__(256)
keccak_ctl_helpers_kernel(bpg::BatchOf<bpg::AuxArgs> batch, uint32_t log_n) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);
  namespace kk = bpg::air::keccak;
  namespace ct = bpg::air::ctl;
  const bpg::AuxArgs& a = batch.a[blockIdx.z];
  const uint32_t n = 1u << log_n, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t first = (i / 24) * 24;
#pragma unroll 1
  for (uint32_t c = 0; c < 2; c++) {
    const uint64_t beta = a.ctl.v[2 * c];
    uint64_t acc = a.trace[(uint64_t)(kk::COL_A + ct::TUPLE_LIMBS - 1) * n + first];
#pragma unroll 1
    for (uint32_t j = ct::TUPLE_LIMBS - 1; j-- > 0;) acc = gl::addc(gl::mulc(acc, beta), a.trace[(uint64_t)(kk::COL_A + j) * n + first]);
    a.aux[(uint64_t)(ct::KECCAK_H + c) * n + i] = acc;
  }
}

// The FILTER columns of the two looked tables -- which rows a table exposes to its lookup (air::ctl) -- are columns of the
// TRACE, written when the witness is generated and committed with it, i.e. before the lookup challenges exist.
// Keccak-f table: g = 1 on the last-round row of permutation p when the looking (sponge) table's row p absorbs a block
// (flag_a + flag_b: the two flag columns of the sponge table's trace, n_flags rows).
__global__ void __launch_bounds__(256)
keccak_lookup_filter_kernel(uint64_t* __restrict__ trace, uint32_t log_n, const uint64_t* __restrict__ flag_a,
                            const uint64_t* __restrict__ flag_b, uint32_t n_flags) {
  const uint32_t n = 1u << log_n, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t perm = i / 24;
  trace[(uint64_t)bpg::air::keccak::COL_G * n + i] = i % 24 == 23 && perm < n_flags && (flag_a[perm] + flag_b[perm]) != 0;
}
// Memory table: g = 1 on the operations the byte-packing table looks up.  A lane owns a packing row that moves a word and
// finds its operation -- (address, timestamp) -- in the memory trace, which is sorted by (address, timestamp), by
// bisection.  pack = the packing table's trace of P rows.  (The column was zeroed by memory_trace_kernel.)
__global__ void __launch_bounds__(256)
memory_lookup_filter_kernel(uint64_t* __restrict__ trace, uint32_t log_n, const uint64_t* __restrict__ pack, uint32_t P) {
  namespace bp = bpg::air::byte_packing;
  namespace mm = bpg::air::memory;
  const uint32_t n = 1u << log_n, r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= P) return;
  uint64_t has = 0;
  for (uint32_t j = 0; j < 32; j++) has += pack[(uint64_t)(bp::COL_LEN + j) * P + r];
  if (!has) return;
  const uint64_t addr = pack[(uint64_t)bp::COL_ADDR * P + r], ts = pack[(uint64_t)bp::COL_TS * P + r];
  const uint64_t *ma = trace + (uint64_t)mm::COL_ADDR * n, *mt = trace + (uint64_t)mm::COL_TS * n;
  uint32_t lo = 0, hi = n;  // first row with (address, timestamp) >= (addr, ts)
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    const uint64_t x = ma[mid], y = mt[mid];
    if (x < addr || (x == addr && y < ts)) lo = mid + 1; else hi = mid;
  }
  if (lo < n && ma[lo] == addr && mt[lo] == ts) trace[(uint64_t)mm::COL_G * n + lo] = 1;
}

// ---------------------------------------------------------------- AIR 8 (plonk, air.hpp): constants, witness, copy products
// The preprocessed columns of the fixed circuit: selectors by the row's place in its group of four, gate constants
// drawn from the circuit's seed, the hash-row selector, and the sigmas sigma_j(w^i) = k_j' w^i' by air::plonk::sigma_of
// for a circuit of the given layout (the list it hashes, the Merkle paths it walks).  grid = (rows/256, 85).
__global__ void __launch_bounds__(256)
plonk_constants_kernel(uint64_t* __restrict__ out, uint32_t log_n, uint64_t seed, const uint64_t* __restrict__ tw_n, bpg::air::plonk::Layout lay) {
  namespace pk = bpg::air::plonk;
  const uint32_t n = 1u << log_n, i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (i >= n) return;
  const uint32_t p = i & 3, a0 = pk::arith_row0(lay);
  uint64_t v;
  if (k == pk::CST_ARITH) v = (i >= a0 && p != 2) || i == pk::ZERO_ROW;   // row 1: the gate that makes the zero wires
  else if (k == pk::CST_SBOX) v = i >= a0 && p == 2;
  else if (k == pk::CST_HASH) v = (i >= pk::HASH_ROW0 && i < pk::HASH_ROW0 + pk::hash_rows(lay.pi_len)) ||
                                  (i >= pk::MERKLE_ROW0 && i < pk::MERKLE_ROW0 + pk::merkle_rows(lay) + pk::leaf_rows(lay));
  else if (k < pk::CST_SIGMA) v = i == pk::ZERO_ROW ? 0 : rnd(seed ^ 0xC0115700C0115700ULL, k, i);
  else {
    uint32_t c2, r2;
    pk::sigma_of(k - pk::CST_SIGMA, i, n, lay, c2, r2);
    v = gl::mulc(gl::pow((uint64_t)7, c2), root_pow(tw_n, log_n, r2));
  }
  out[(uint64_t)k * n + i] = v;
}
// The witness: one lane per (group of four rows, slot): the slot's four routed wires in the group's rows and, for the
// first eleven slots, the five advice wires of their S-box unit.  Free wires are rnd(seed, column, row) as in the
// synthetic witness; what a slot needs of its neighbour (b_s of row 4g + 1 is d_(s+1) of row 4g) it recomputes from the
// seed.  No lane keeps more than a handful of values: nothing lives in scratch.  grid.x covers (n / 4) x 20 lanes,
// groups fastest.
__global__ void __launch_bounds__(256)
plonk_trace_kernel(bpg::BatchOf<bpg::PlonkTraceArgs> batch, uint32_t log_n) {
  namespace pk = bpg::air::plonk;
  if (gridDim.x * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);
  const uint32_t n = 1u << log_n, groups = n >> 2, id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= groups * pk::N_SLOTS) return;
  const uint32_t g = id % groups, s = id / groups, r0 = 4 * g;
  uint64_t* __restrict__ t = batch.a[blockIdx.z].trace;
  const uint64_t* __restrict__ cs = batch.a[blockIdx.z].consts;
  const uint64_t seed = batch.a[blockIdx.z].seed;
  const uint32_t sn = (s + 1) % pk::N_SLOTS;  // the neighbour slot whose output this slot's b input copies
  const uint64_t pub_s = s < 4 ? batch.a[blockIdx.z].pub[s] : 0, pub_n = sn < 4 ? batch.a[blockIdx.z].pub[sn] : 0;
  const bool unit = s < pk::N_SBOX;
  const uint32_t u = pk::COL_SBOX + 5 * s;  // the slot's S-box unit (slots 0..10)
#define PUT(col, row, v) t[(uint64_t)(col) * n + (row)] = (v)
#define CST(k, row) cs[(uint64_t)(k) * n + (row)]
  if (g == 0) {  // the public-input row, the zero row (its d wires are 0: an arithmetic gate with c0 = c1 = 0) and two no-op rows
    for (uint32_t r = 0; r < 4; r++) {
      for (uint32_t w = 0; w < 4; w++) PUT(4 * s + w, r, r == pk::ZERO_ROW && w == 3 ? 0 : rnd(seed, 4 * s + w, r));
      if (unit)
        for (uint32_t w = 0; w < 5; w++) PUT(u + w, r, rnd(seed, u + w, r));
    }
    if (s == 0)
      for (uint32_t j = 0; j < 4; j++) PUT(j, 0, batch.a[blockIdx.z].pub[j]);
    return;
  }
  const uint32_t a0 = batch.a[blockIdx.z].arith_row0;
  if (r0 < a0) return;  // rows 4 .. a0 - 1: the hash rows and the Merkle rows, written by plonk_hash_rows_kernel
  const bool first_group = r0 == a0;  // its c inputs are the public inputs
  if (unit)  // the advice wires of the three arithmetic rows are free
    for (uint32_t p = 0; p < 4; p++)
      if (p != 2)
        for (uint32_t w = 0; w < 5; w++) PUT(u + w, r0 + p, rnd(seed, u + w, r0 + p));
  // row 4g: inputs free (c_j of the first computing row = public input j)
  const uint64_t c0 = CST(pk::CST_C0, r0), c1 = CST(pk::CST_C1, r0);
  const uint64_t av = rnd(seed, 4 * s, r0), bv = rnd(seed, 4 * s + 1, r0);
  const uint64_t cv = first_group && s < 4 ? pub_s : rnd(seed, 4 * s + 2, r0);
  const uint64_t d0 = gl::addc(gl::mulc(c0, gl::mulc(av, bv)), gl::mulc(c1, cv));
  const uint64_t cn = first_group && sn < 4 ? pub_n : rnd(seed, 4 * sn + 2, r0);
  const uint64_t d0n = gl::addc(gl::mulc(c0, gl::mulc(rnd(seed, 4 * sn, r0), rnd(seed, 4 * sn + 1, r0))), gl::mulc(c1, cn));
  PUT(4 * s, r0, av); PUT(4 * s + 1, r0, bv); PUT(4 * s + 2, r0, cv); PUT(4 * s + 3, r0, d0);
  // row 4g + 1: a_s = d_s, b_s = d_(s+1), c_s = c_s of the row above
  const uint64_t d1 = gl::addc(gl::mulc(CST(pk::CST_C0, r0 + 1), gl::mulc(d0, d0n)), gl::mulc(CST(pk::CST_C1, r0 + 1), cv));
  PUT(4 * s, r0 + 1, d0); PUT(4 * s + 1, r0 + 1, d0n); PUT(4 * s + 2, r0 + 1, cv); PUT(4 * s + 3, r0 + 1, d1);
  // row 4g + 2: the S-box units take d_i of the row above to the 7th power
  uint64_t a3 = rnd(seed, 4 * s, r0 + 3);
  if (unit) {
    const uint64_t x = d1, x2 = gl::mulc(x, x), x4 = gl::mulc(x2, x2), x6 = gl::mulc(x4, x2), x7 = gl::mulc(x6, x);
    PUT(u, r0 + 2, x); PUT(u + 1, r0 + 2, x2); PUT(u + 2, r0 + 2, x4); PUT(u + 3, r0 + 2, x6); PUT(u + 4, r0 + 2, x7);
    PUT(4 * s, r0 + 2, x); PUT(4 * s + 3, r0 + 2, x7);
    a3 = x7;
  } else {
    PUT(4 * s, r0 + 2, rnd(seed, 4 * s, r0 + 2)); PUT(4 * s + 3, r0 + 2, rnd(seed, 4 * s + 3, r0 + 2));
  }
  PUT(4 * s + 1, r0 + 2, rnd(seed, 4 * s + 1, r0 + 2)); PUT(4 * s + 2, r0 + 2, rnd(seed, 4 * s + 2, r0 + 2));
  // row 4g + 3: a_i = the S-box outputs, the rest free
  const uint64_t b3 = rnd(seed, 4 * s + 1, r0 + 3), c3 = rnd(seed, 4 * s + 2, r0 + 3);
  const uint64_t d3 = gl::addc(gl::mulc(CST(pk::CST_C0, r0 + 3), gl::mulc(a3, b3)), gl::mulc(CST(pk::CST_C1, r0 + 3), c3));
  PUT(4 * s, r0 + 3, a3); PUT(4 * s + 1, r0 + 3, b3); PUT(4 * s + 2, r0 + 3, c3); PUT(4 * s + 3, r0 + 3, d3);
#undef PUT
#undef CST
}
// The Poseidon rows (4 .. arith_row0 - 1) of the witness: row 4 + h takes its wires from the rows the host made -- h <
// n_hash_rows while it hashed the public-input list (bpg::poseidon_hash_rows: the list is a few dozen words and the host
// has just hashed it anyway -- one lane chaining six permutations would cost 0.3 ms of a proof's critical path,
// profiles/r5_transcript_latency.txt), 8 <= h < 8 + n_merkle_rows while it walked the children's Merkle paths
// (bpg::poseidon_merkle_rows); every other row of the region is free.  grid = (1, rows of the region, proofs) x 135 lanes.
__global__ void __launch_bounds__(256)
plonk_hash_rows_kernel(bpg::BatchOf<bpg::PlonkTraceArgs> batch, uint32_t log_n) {
  namespace pk = bpg::air::plonk;
  __builtin_amdgcn_s_setprio(3);
  const bpg::PlonkTraceArgs& a = batch.a[blockIdx.z];
  const uint32_t n = 1u << log_n, c = threadIdx.x, h = blockIdx.y, row = pk::HASH_ROW0 + h;
  if (c >= pk::N_COLS || row >= a.arith_row0) return;
  const bool given = h < a.n_hash_rows || (h >= pk::HASH_ROWS_MAX && h < pk::HASH_ROWS_MAX + a.n_merkle_rows);
  a.trace[(uint64_t)c * n + row] = given ? a.hash_rows[(uint64_t)h * pk::H_WIRES + c] : rnd(a.seed, c, row);
}
// Copy products, step 1 of 3: per row and challenge set the ten chunk ratios num_k / den_k (one inversion per row:
// the denominators are inverted together) as cumulative products P_k = prod_{k' <= k} num / den: P_1..P_9 into the
// partial-product columns, the row's total P_10 into the Z column.  grid = (rows/256, 2, proofs).
__global__ void __launch_bounds__(256)
plonk_chunk_ratios_kernel(bpg::BatchOf<bpg::AuxArgs> batch, uint32_t log_n, const uint64_t* __restrict__ tw_n) {
  namespace pk = bpg::air::plonk;
  namespace ct = bpg::air::ctl;
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);
  const bpg::AuxArgs& a = batch.a[blockIdx.z];
  const uint32_t n = 1u << log_n, i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (i >= n) return;
  const uint64_t beta = a.ctl.v[2 * c], gamma = a.ctl.v[2 * c + 1];
  uint64_t bkx = gl::mulc(beta, root_pow(tw_n, log_n, i));
  uint64_t num[ct::PLONK_CHUNKS], den[ct::PLONK_CHUNKS];
#pragma unroll 1
  for (uint32_t k = 0; k < ct::PLONK_CHUNKS; k++) {
    uint64_t nu = 1, de = 1;
#pragma unroll 1
    for (uint32_t j = ct::PLONK_CHUNK * k; j < ct::PLONK_CHUNK * (k + 1); j++) {
      const uint64_t w = gl::addc(a.trace[(uint64_t)j * n + i], gamma);
      nu = gl::mulc(nu, gl::addc(w, bkx));
      de = gl::mulc(de, gl::addc(w, gl::mulc(beta, a.consts[(uint64_t)(pk::CST_SIGMA + j) * n + i])));
      bkx = gl::mulc(bkx, 7);
    }
    num[k] = nu; den[k] = de;
  }
  // 1 / den_k for all k with one inversion: prefix products, invert the last, peel backwards
  uint64_t pre[ct::PLONK_CHUNKS];
  pre[0] = den[0];
#pragma unroll
  for (uint32_t k = 1; k < ct::PLONK_CHUNKS; k++) pre[k] = gl::mulc(pre[k - 1], den[k]);
  uint64_t inv = gl::inv(pre[ct::PLONK_CHUNKS - 1]);
#pragma unroll
  for (uint32_t k = ct::PLONK_CHUNKS; k-- > 0;) {
    const uint64_t dk_inv = k ? gl::mulc(inv, pre[k - 1]) : inv;
    inv = gl::mulc(inv, den[k]);
    num[k] = gl::mulc(num[k], dk_inv);  // the ratio of chunk k
  }
  uint64_t P = 1;
  uint64_t* aux = a.aux + (uint64_t)(10 * c) * n + i;
#pragma unroll
  for (uint32_t k = 0; k < ct::PLONK_CHUNKS; k++) {
    P = gl::mulc(P, num[k]);
    aux[(uint64_t)(k + 1 < ct::PLONK_CHUNKS ? k + 1 : 0) * n] = P;  // P_(k+1): columns 1..9, the total in column 0
  }
}
// step 3 of 3 (step 2, the suffix products of the row totals in the two Z columns, is aux_suffix_product_kernel):
// partial product k of row i = Z(i + 1) P_k(i), the wrap from the last row to the first included.  grid = (rows/256, 18, proofs).
__global__ void __launch_bounds__(256) plonk_partial_products_kernel(bpg::BatchOf<bpg::AuxArgs> batch, uint32_t log_n) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);
  const bpg::AuxArgs& a = batch.a[blockIdx.z];
  const uint32_t n = 1u << log_n, i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y / 9, k = 1 + blockIdx.y % 9;
  if (i >= n) return;
  const uint64_t z_next = a.aux[(uint64_t)(10 * c) * n + ((i + 1) & (n - 1))];
  uint64_t* p = a.aux + (uint64_t)(10 * c + k) * n + i;
  *p = gl::mulc(*p, z_next);
}

// ---------------------------------------------------------------- K5 quotient
// One kernel for every AIR (air.hpp): a lane owns one point of the LDE coset, evaluates the units its workgroup
// row was given and folds the constraints with the alpha powers IN REGISTERS:
//   acc_j = sum_i c_i * alpha_j^(T-1-i)      (= starky's acc = acc * alpha + c over the whole list),
// kept as unreduced dot-product accumulators (gl::DotAcc: 8 VALU per term, one reduction at the end).  A table
// tall enough to fill the chip takes ONE pass (grid.y = 1): the quotient value is written straight away and no
// partial sums ever reach HBM.  Short tables spread their units over grid.y workgroup rows; the partial sums of
// the rows simply add (every term carries its absolute power), which quotient_sum_kernel does.
struct RowPoint {
  uint64_t z_last, l_first, l_last, x;
};
__device__ __forceinline__ RowPoint row_point(const bpg::QuotArgs& q, uint32_t t, uint32_t m) {
  // per-coset constants from the table alpha_table_kernel left behind the alpha powers (t is a per-lane value:
  // indexing the kernel-argument arrays with it would pull all 48 words into SGPRs)
  const uint64_t* coset = q.apow + 2 * (size_t)q.n_constraints;
  const uint64_t x = gl::mulc(coset[t], root_pow(q.tw_n, q.log_n, m));
  const uint64_t zh = coset[16 + t];
  RowPoint p;
  p.x = x;
  p.z_last = gl::subc(x, q.g_inv);
  const uint64_t zn = gl::mulc(zh, q.n_inv);
  // both Lagrange denominators with one inversion (x is off the subgroup: neither is zero)
  const uint64_t df = gl::subc(x, 1), dl = gl::subc(gl::mulc(q.g, x), 1);
  const uint64_t both = gl::mulc(zn, gl::inv(gl::mulc(df, dl)));
  p.l_first = gl::mulc(both, dl);
  p.l_last = gl::mulc(both, df);
  return p;
}
struct DevRow {  // one point of the coset: column-major matrices, lanes = consecutive rows (coalesced)
  const uint64_t *trace, *aux_, *cst_;
  uint64_t ts, as, cs, pos, pos_next;
  uint64_t xv;          // the point itself
  const uint64_t* pub_;  // the table's public inputs (kernel arguments)
  __device__ __forceinline__ uint64_t x() const { return xv; }
  __device__ __forceinline__ uint64_t pub(uint32_t j) const { return pub_[j]; }
  __device__ __forceinline__ uint64_t loc(uint32_t c) const { return trace[(uint64_t)c * ts + pos]; }
  __device__ __forceinline__ uint64_t nxt(uint32_t c) const { return trace[(uint64_t)c * ts + pos_next]; }
  __device__ __forceinline__ uint64_t cst(uint32_t k) const { return cst_[(uint64_t)k * cs + pos]; }
  __device__ __forceinline__ uint64_t aux(uint32_t k) const { return aux_[(uint64_t)k * as + pos]; }
  __device__ __forceinline__ uint64_t aux_nxt(uint32_t k) const { return aux_[(uint64_t)k * as + pos_next]; }
};
struct DevEmit {  // the constraint consumer: two constraints (x two challenges) per dot_mad4
  const uint64_t* apow;  // [2][T]: alpha_j^e (wave-uniform reads)
  uint32_t T;
  RowPoint rp;
  gl::DotAcc acc[4];  // 0, 1: challenge 0 / 1 of the first constraint of a pair; 2, 3: of the second
  uint64_t pend_v;
  uint32_t pend_e;
  bool has;
  __device__ __forceinline__ void push(uint32_t idx, uint64_t v) {
    const uint32_t e = T - 1 - idx;
    if (!has) {
      pend_v = v; pend_e = e; has = true;
      return;
    }
    const uint64_t a[4] = {pend_v, pend_v, v, v};
    const uint64_t w[4] = {apow[pend_e], apow[T + pend_e], apow[e], apow[T + e]};
    gl::dot_mad4(acc, a, w);
    has = false;
  }
  __device__ __forceinline__ void all(uint32_t idx, uint64_t v) { push(idx, v); }
  __device__ __forceinline__ void transition(uint32_t idx, uint64_t v) { push(idx, gl::mulc(v, rp.z_last)); }
  __device__ __forceinline__ void first(uint32_t idx, uint64_t v) { push(idx, gl::mulc(v, rp.l_first)); }
  __device__ __forceinline__ void last(uint32_t idx, uint64_t v) { push(idx, gl::mulc(v, rp.l_last)); }
  __device__ __forceinline__ uint64_t result(int j) {
    if (has) {
      const uint64_t a[4] = {pend_v, pend_v, 0, 0};
      const uint64_t w[4] = {apow[pend_e], apow[T + pend_e], 0, 0};
      gl::dot_mad4(acc, a, w);
      has = false;
    }
    return gl::addc(gl::dot_reduce(acc[j]), gl::dot_reduce(acc[2 + j]));
  }
};
// grid = (rows / 256, workgroup rows); workgroup row y evaluates units [y * units_per_wg, ...) of the list
// "AIR units, then CTL units".
// The Keccak-f evaluator wants 193 VGPRs (two waves per SIMD); held to 168 (three waves, 26 registers in scratch) it is
// 10 % faster alone on the chip (700 -> 628 us at 2^14 rows; four waves at 128 VGPRs: 795 us).
#ifndef BPG_PLONK_HASH_WAVES
#define BPG_PLONK_HASH_WAVES 2
#endif
template <uint32_t AIR>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(AIR == bpg::air::KECCAK_F ? 3 : 1)))
quotient_air_kernel(bpg::BatchOf<bpg::QuotArgs> batch) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  const bpg::QuotArgs& q = batch.a[blockIdx.z];
  const uint64_t rows = (uint64_t)1 << (q.log_n + q.rate_bits);
  const uint64_t pos = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (pos >= rows) return;
  const uint32_t n = 1u << q.log_n;
  const uint32_t t = (uint32_t)(pos >> q.log_n), m = (uint32_t)(pos & (n - 1));
  DevEmit out{q.apow, q.n_constraints, row_point(q, t, m),
              {gl::dot_zero(), gl::dot_zero(), gl::dot_zero(), gl::dot_zero()}, 0, 0, false};
  DevRow row{q.trace_lde, q.aux_lde, q.const_lde, q.trace_stride, q.aux_stride, q.const_stride, pos,
             ((uint64_t)t << q.log_n) | ((m + 1) & (n - 1)), out.rp.x, q.ctl.pub};
  const bpg::air::Shape shape{AIR, q.n_cols, q.n_const, q.deg_pow};
  const uint32_t n_units = q.n_air_units + q.n_ctl_units;
  const uint32_t u0 = blockIdx.y * q.units_per_wg, u1 = min(u0 + q.units_per_wg, n_units);
#pragma unroll 1
  for (uint32_t u = u0; u < u1; u++) {
    if (u < q.n_air_units) {
      if constexpr (AIR == bpg::air::KECCAK_F) bpg::air::keccak::eval_unit<uint64_t>(u, row, out);
      else if constexpr (AIR == bpg::air::LOGIC) bpg::air::logic::eval_unit<uint64_t>(u, q.n_air_constraints, q.ctl.v, row, out);
      else if constexpr (AIR == bpg::air::MEMORY) bpg::air::memory::eval_unit<uint64_t>(row, out);
      else if constexpr (AIR == bpg::air::ARITHMETIC) bpg::air::arithmetic::eval_unit<uint64_t>(u, row, out);
      else if constexpr (AIR == bpg::air::BYTE_PACKING) bpg::air::byte_packing::eval_unit<uint64_t>(u, row, out);
      else if constexpr (AIR == bpg::air::KECCAK_SPONGE) bpg::air::keccak_sponge::eval_unit<uint64_t>(u, q.n_air_constraints, q.ctl.v, row, out);
      else if constexpr (AIR == bpg::air::ARITHMETIC_MUL) bpg::air::arithmetic_mul::eval_unit<uint64_t>(u, row, out);
      else if constexpr (AIR == bpg::air::PLONK) bpg::air::plonk::eval_chunk_unit<uint64_t>(u, q.n_air_constraints, q.ctl.v, row, out);  // (unit 10: quotient_plonk_hash_kernel)
      else bpg::air::synthetic::eval_unit<uint64_t>(shape, u, row, out);
    } else {
      const uint32_t k0 = (u - q.n_air_units) * q.aux_per_unit, k1 = min(k0 + q.aux_per_unit, q.n_aux);
      const bpg::air::Shape cs{AIR, q.n_cols, q.n_const, q.deg_pow};  // AIR as a constant: the other tables' lookups fold away
      bpg::air::ctl::eval<uint64_t>(cs, q.n_air_constraints, k0, k1, q.ctl.v, row, out);
    }
  }
  const uint64_t r0 = out.result(0), r1 = out.result(1);
  if (gridDim.y == 1 && q.side_rows == 0) {
    const uint64_t zh_inv = q.apow[2 * (size_t)q.n_constraints + 32 + t];
    q.qvals[pos] = gl::mulc(r0, zh_inv);
    q.qvals[rows + pos] = gl::mulc(r1, zh_inv);
  } else {
    q.partial[((uint64_t)blockIdx.y * 2) * rows + pos] = r0;
    q.partial[((uint64_t)blockIdx.y * 2 + 1) * rows + pos] = r1;
  }
}
// AIR 8's Poseidon gate: 118 local checks of one permutation's wires per row, folded bare and multiplied by the
// selector once; its sums are row `wg_row` of `partial`.  About ten times the instructions of a chunk unit and another
// register budget (the twelve-word state and the layer's accumulators), hence its own kernel: the chunk units keep three
// waves per SIMD.   grid = (rows / 256, 1, batch)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BPG_PLONK_HASH_WAVES)))
quotient_plonk_hash_kernel(bpg::BatchOf<bpg::QuotArgs> batch, uint32_t wg_row) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  const bpg::QuotArgs& q = batch.a[blockIdx.z];
  const uint64_t rows = (uint64_t)1 << (q.log_n + q.rate_bits);
  const uint64_t pos = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (pos >= rows) return;
  const uint32_t n = 1u << q.log_n;
  const uint32_t t = (uint32_t)(pos >> q.log_n), m = (uint32_t)(pos & (n - 1));
  DevEmit out{q.apow, q.n_constraints, row_point(q, t, m),
              {gl::dot_zero(), gl::dot_zero(), gl::dot_zero(), gl::dot_zero()}, 0, 0, false};
  DevRow row{q.trace_lde, q.aux_lde, q.const_lde, q.trace_stride, q.aux_stride, q.const_stride, pos,
             ((uint64_t)t << q.log_n) | ((m + 1) & (n - 1)), out.rp.x, q.ctl.pub};
  bpg::air::plonk::eval_hash_unit<uint64_t, DevRow, DevEmit, false>(row, out);
  const uint64_t qh = row.cst(bpg::air::plonk::CST_HASH);
  q.partial[((uint64_t)wg_row * 2) * rows + pos] = gl::mulc(out.result(0), qh);
  q.partial[((uint64_t)wg_row * 2 + 1) * rows + pos] = gl::mulc(out.result(1), qh);
}
// qvals = (sum of the workgroup rows' partial sums) / Z_H.   grid = (rows / 256, 2 challenges)
__global__ void __launch_bounds__(256) quotient_sum_kernel(bpg::BatchOf<bpg::QuotArgs> batch, uint32_t n_wg_rows) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  const bpg::QuotArgs& q = batch.a[blockIdx.z];
  const uint64_t rows = (uint64_t)1 << (q.log_n + q.rate_bits);
  const uint64_t pos = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (pos >= rows) return;
  const uint32_t j = blockIdx.y, t = (uint32_t)(pos >> q.log_n);
  uint64_t acc = 0;
  for (uint32_t c = 0; c < n_wg_rows; c++) acc = gl::addc(acc, q.partial[((uint64_t)c * 2 + j) * rows + pos]);
  q.qvals[(uint64_t)j * rows + pos] = gl::mulc(acc, q.apow[2 * (size_t)q.n_constraints + 32 + t]);
}
// apow[j * T + e] = alpha_j^e, then the per-coset constants: g_t[16], zh_t[16], zh_inv_t[16]
struct CosetTab {
  uint64_t v[48];
};
struct AlphaTabArgs {
  uint64_t* out;
  uint64_t a0, a1;
};
__global__ void __launch_bounds__(256)
alpha_table_kernel(bpg::BatchOf<AlphaTabArgs> batch, uint32_t T, CosetTab ct) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  uint64_t* __restrict__ out = batch.a[blockIdx.z].out;
  const uint64_t a0 = batch.a[blockIdx.z].a0, a1 = batch.a[blockIdx.z].a1;
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.y == 0 && e < 48) out[2 * (uint64_t)T + e] = ct.v[e];
  if (e >= T) return;
  out[(uint64_t)blockIdx.y * T + e] = gl::pow(blockIdx.y ? a1 : a0, e);
}
// After the per-coset inverse NTT: E_t[pos] (bit-reversed n0).  c_{n0 + n*n1} =
// (s^n)^(-n1) / 2^r * sum_t w_{2^r}^(-t n1) * E_t[pos] * g_t^(-n0).   grid = (n/256, 2 challenges)
__global__ void __launch_bounds__(256) quotient_chunks_kernel(bpg::BatchOf<bpg::ChunkArgs> batch) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  const bpg::ChunkArgs& c = batch.a[blockIdx.z];
  const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x, n = 1u << c.log_n;
  if (pos >= n) return;
  const uint32_t j = blockIdx.y, R = 1u << c.rate_bits;
  const uint64_t* e = c.e + (uint64_t)j * ((uint64_t)n << c.rate_bits);
  uint64_t d[16];
  for (uint32_t t = 0; t < R; t++) d[t] = gl::mulc(e[(uint64_t)t * n + pos], c.inv_scale[(uint64_t)t * n + pos]);
  for (uint32_t n1 = 0; n1 < R; n1++) {
    uint64_t acc = 0;
    for (uint32_t t = 0; t < R; t++) acc = gl::addc(acc, gl::mulc(d[t], c.wr_inv_pow[(t * n1) & (R - 1)]));
    c.out[((uint64_t)j * R + n1) * c.out_stride + pos] = gl::mulc(acc, c.chunk_scale[n1]);
  }
}

// ---------------------------------------------------------------- K8 openings
// pw[0..n) = zeta^bitrev(pos) (c0 plane), pw[n..2n) c1 plane; same for the second point at 2n.
__global__ void __launch_bounds__(256)
power_vector_kernel(bpg::BatchOf<bpg::PowerVecArgs> batch, uint32_t log_n) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x, n = 1u << log_n;
  if (pos >= n) return;
  const uint32_t e = gl::bitrev(pos, log_n);
  const Ext z = batch.a[blockIdx.z].z[blockIdx.y];
  const Ext r = gl::pow(z, e);
  uint64_t* o = batch.a[blockIdx.z].out + (uint64_t)blockIdx.y * 2 * n;
  o[pos] = r.c0;
  o[n + pos] = r.c1;
}
__global__ void __launch_bounds__(256) alpha_pows_kernel(bpg::BatchOf<bpg::AlphaPowArgs> batch, uint32_t count) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  uint64_t* __restrict__ out = batch.a[blockIdx.z].out;
  const Ext alpha = batch.a[blockIdx.z].alpha;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  const Ext r = gl::pow(alpha, j);
  out[2 * j] = r.c0;
  out[2 * j + 1] = r.c1;
}
// One workgroup per coefficient column: out[col] = (sum c*p0.c0, sum c*p0.c1, sum c*p1.c0, sum c*p1.c1)
__device__ __forceinline__ void
openings_column(const uint64_t* __restrict__ c, uint32_t log_n, const uint64_t* __restrict__ pw, uint32_t n_points,
                uint64_t* __restrict__ out4) {
  __shared__ uint64_t red[4][256];
  const uint32_t n = 1u << log_n;
  // unreduced accumulation (gl::DotAcc): 8 VALU per product, one reduction per lane at the end
  gl::DotAcc dacc[4] = {gl::dot_zero(), gl::dot_zero(), gl::dot_zero(), gl::dot_zero()};
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint64_t v = c[i];
    const uint64_t vv[4] = {v, v, v, v};
    uint64_t w[4] = {pw[i], pw[n + i], 0, 0};
    if (n_points > 1) {
      w[2] = pw[2 * n + i];
      w[3] = pw[3 * n + i];
    }
    gl::dot_mad4(dacc, vv, w);
  }
  uint64_t acc[4];
#pragma unroll
  for (int k = 0; k < 4; k++) acc[k] = gl::dot_reduce(dacc[k]);
  for (int k = 0; k < 4; k++) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (uint32_t s = blockDim.x / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s)
      for (int k = 0; k < 4; k++) red[k][threadIdx.x] = gl::addc(red[k][threadIdx.x], red[k][threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x < 4) out4[threadIdx.x] = red[threadIdx.x][0];
}
__global__ void __launch_bounds__(256)
openings_kernel(const uint64_t* __restrict__ coeffs, uint64_t stride, uint32_t log_n,
                const uint64_t* __restrict__ pw, uint32_t n_points, uint64_t* __restrict__ out) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  openings_column(coeffs + blockIdx.x * stride, log_n, pw, n_points, out + (uint64_t)blockIdx.x * 4);
}
// every opening set of a proof (constants, trace, aux, quotient at zeta [/ g zeta], aux at 1) in ONE launch: the
// five launches it replaces sat one behind the other on the proof's critical path
__global__ void __launch_bounds__(256) openings_multi_kernel(bpg::BatchOf<bpg::OpenMulti> batch) {
  if (gridDim.x * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);
  const bpg::OpenMulti& m = batch.a[blockIdx.z];
  uint32_t s = 0;
#pragma unroll
  for (int k = 1; k < 5; k++)
    if (k < (int)m.n_segs && blockIdx.x >= m.first_col[k]) s = k;
  const uint32_t col = blockIdx.x - m.first_col[s];
  openings_column(m.coeffs[s] + (uint64_t)col * m.stride, m.log_n, m.pw[s], m.n_points[s], m.out[s] + (uint64_t)col * 4);
}

// ---------------------------------------------------------------- K6a FRI combine
// Coefficient-space alpha reduction.  For a block of columns of ONE oracle, accumulates into up to
// three batch polynomials:  G_b[pos] += sum_col alpha^(e_b + col) * coeff[col][pos].
// grid = (n/256, column chunks).  partial layout: [chunk][b][2][n]
// columns [c0, c1) of one oracle into the six unreduced accumulators (3 batches x 2 extension components, two groups)
__device__ __forceinline__ void fri_combine_accumulate(const bpg::CombineArgs& a, uint32_t c0, uint32_t c1, uint32_t pos,
                                                       gl::DotAcc (&d01)[4], gl::DotAcc (&d2)[4]) {
  const bool on0 = a.exp_base[0] >= 0, on1 = a.exp_base[1] >= 0, on2 = a.exp_base[2] >= 0;
  for (uint32_t c = c0; c < c1; c++) {
    const uint64_t v = a.coeffs[(uint64_t)c * a.stride + pos];
    const uint64_t vv[4] = {v, v, v, v};
    if (on0 || on1) {
      uint64_t w[4] = {0, 0, 0, 0};
      if (on0) {
        const uint64_t* ap = a.alpha_pows + 2 * ((uint64_t)a.exp_base[0] + c);
        w[0] = ap[0];
        w[1] = ap[1];
      }
      if (on1) {
        const uint64_t* ap = a.alpha_pows + 2 * ((uint64_t)a.exp_base[1] + c);
        w[2] = ap[0];
        w[3] = ap[1];
      }
      gl::dot_mad4(d01, vv, w);
    }
    if (on2) {
      const uint64_t* ap = a.alpha_pows + 2 * ((uint64_t)a.exp_base[2] + c);
      const uint64_t w[4] = {ap[0], ap[1], 0, 0};
      gl::dot_mad4(d2, vv, w);
    }
  }
}
__device__ __forceinline__ void fri_combine_store(gl::DotAcc (&d01)[4], gl::DotAcc (&d2)[4], uint64_t* out, uint32_t n, uint32_t pos) {
  const Ext acc[3] = {Ext{gl::dot_reduce(d01[0]), gl::dot_reduce(d01[1])}, Ext{gl::dot_reduce(d01[2]), gl::dot_reduce(d01[3])},
                      Ext{gl::dot_reduce(d2[0]), gl::dot_reduce(d2[1])}};
#pragma unroll
  for (int b = 0; b < 3; b++) {
    out[(2 * b) * (uint64_t)n + pos] = acc[b].c0;
    out[(2 * b + 1) * (uint64_t)n + pos] = acc[b].c1;
  }
}
__device__ __forceinline__ void fri_combine_partial_body(const bpg::CombineArgs& a, uint32_t chunk) {
  const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x, n = 1u << a.log_n;
  if (pos >= n) return;
  const uint32_t c0 = chunk * a.cols_per_chunk, c1 = min(c0 + a.cols_per_chunk, a.n_cols);
  gl::DotAcc d01[4] = {gl::dot_zero(), gl::dot_zero(), gl::dot_zero(), gl::dot_zero()};
  gl::DotAcc d2[4] = {gl::dot_zero(), gl::dot_zero(), gl::dot_zero(), gl::dot_zero()};
  fri_combine_accumulate(a, c0, c1, pos, d01, d2);
  fri_combine_store(d01, d2, a.partial + (uint64_t)(a.chunk_base + chunk) * 6 * n, n, pos);
}
// Every column of every oracle in one pass, straight into g[6][n]: no partial sums, no reduce launch.  For a
// device that several provers share (nothing needs filling); the sums are exact, so the bytes are the same.
struct GPtrs {
  uint64_t* g[bpg::MAX_BATCH];
};
__global__ void __launch_bounds__(256) fri_combine_all_kernel(bpg::BatchOf<bpg::CombineMulti> batch, GPtrs gp) {
  if (gridDim.x * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);
  const bpg::CombineMulti& m = batch.a[blockIdx.z];
  uint64_t* __restrict__ g = gp.g[blockIdx.z];
  const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x, n = 1u << m.a[0].log_n;
  if (pos >= n) return;
  gl::DotAcc d01[4] = {gl::dot_zero(), gl::dot_zero(), gl::dot_zero(), gl::dot_zero()};
  gl::DotAcc d2[4] = {gl::dot_zero(), gl::dot_zero(), gl::dot_zero(), gl::dot_zero()};
  for (uint32_t o = 0; o < m.n_oracles; o++) fri_combine_accumulate(m.a[o], 0, m.a[o].n_cols, pos, d01, d2);
  fri_combine_store(d01, d2, g, n, pos);
}
__global__ void __launch_bounds__(256) fri_combine_partial_kernel(bpg::CombineArgs a) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  fri_combine_partial_body(a, blockIdx.y);
}
// all oracles of a proof (constants, trace, aux, quotient) in one launch: grid.y runs over the chunks of all of them
__global__ void __launch_bounds__(256) fri_combine_partial_multi_kernel(bpg::BatchOf<bpg::CombineMulti> batch) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);
  const bpg::CombineMulti& m = batch.a[blockIdx.z];
  uint32_t o = 0;
#pragma unroll
  for (int k = 1; k < 4; k++)
    if (k < (int)m.n_oracles && blockIdx.y >= m.a[k].chunk_base) o = k;
  fri_combine_partial_body(m.a[o], blockIdx.y - m.a[o].chunk_base);
}
// g[6][n] = sum over chunks.  grid = (n/256, 6)
__global__ void __launch_bounds__(256)
fri_combine_reduce_kernel(bpg::BatchOf<bpg::CombineReduceArgs> batch, uint32_t n_chunks, uint32_t log_n) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  const uint64_t* __restrict__ partial = batch.a[blockIdx.z].partial;
  uint64_t* __restrict__ g = batch.a[blockIdx.z].g;
  const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x, n = 1u << log_n;
  if (pos >= n) return;
  uint64_t acc = 0;
  for (uint32_t c = 0; c < n_chunks; c++) acc = gl::addc(acc, partial[((uint64_t)c * 6 + blockIdx.y) * n + pos]);
  g[(uint64_t)blockIdx.y * n + pos] = acc;
}
// Layer-0 FRI values: V(x) = sum_b alpha^(e_b) * (G_b(x) - y_b) / (x - z_b), x on the LDE coset.
// glde: [6][rows] coset-major; out: AoS ext [rows].
__global__ void __launch_bounds__(256) fri_quotient_values_kernel(bpg::BatchOf<bpg::FriInitArgs> batch) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  const bpg::FriInitArgs& a = batch.a[blockIdx.z];
  const uint64_t rows = (uint64_t)1 << (a.log_n + a.rate_bits);
  const uint64_t pos = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (pos >= rows) return;
  const uint32_t t = (uint32_t)(pos >> a.log_n), m = (uint32_t)(pos & ((1u << a.log_n) - 1));
  const uint64_t x = gl::mulc(a.g_t[t], root_pow(a.tw_n, a.log_n, m));
  // 1/(x - z_b) for the three points with ONE inversion (x is on the coset, z_b is a transcript
  // challenge in the extension: the denominators are nonzero)
  const Ext d0 = gl::sub(gl::ext(x), a.z[0]), d1 = gl::sub(gl::ext(x), a.z[1]), d2 = gl::sub(gl::ext(x), a.z[2]);
  const Ext d01 = gl::mul(d0, d1);
  const Ext all_inv = gl::inv(gl::mul(d01, d2));
  const Ext i2 = gl::mul(all_inv, d01), i01 = gl::mul(all_inv, d2);
  const Ext den_inv[3] = {gl::mul(i01, d1), gl::mul(i01, d0), i2};
  Ext sum = gl::ext(0);
#pragma unroll
  for (int b = 0; b < 3; b++) {
    const Ext gv{a.glde[(2 * b) * rows + pos], a.glde[(2 * b + 1) * rows + pos]};
    const Ext num = gl::sub(gv, a.y[b]);
    sum = gl::add(sum, gl::mul(a.alpha_shift[b], gl::mul(num, den_inv[b])));
  }
  a.out[2 * pos] = sum.c0;
  a.out[2 * pos + 1] = sum.c1;
}

// ---------------------------------------------------------------- K6b FRI layer: leaves + fold
// Layer of m = n_l << r ext values, coset-major.  Lane (t, m0), m0 < n_l/a, owns the a points
// x0 * w_a^j', j' = 0..a-1 at positions t*n_l + m0 + j'*(n_l/a).  Its Merkle leaf (upstream:
// bit-reversed values chunked by arity) is those values in bitrev_a(j) order, leaf index
// bitrev(t + 2^r*m0).
__global__ void __launch_bounds__(256) fri_layer_leaf_kernel(bpg::BatchOf<bpg::FriLayerArgs> batch) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  const bpg::FriLayerArgs& a = batch.a[blockIdx.z];
  const uint32_t log_q = a.log_nl - a.arity_bits;  // log2(n_l / arity)
  const uint64_t n_leaves = (uint64_t)1 << (log_q + a.rate_bits);
  const uint64_t id = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (id >= n_leaves) return;
  const uint32_t t = (uint32_t)(id >> log_q), m0 = (uint32_t)(id & ((1u << log_q) - 1));
  const uint64_t base = ((uint64_t)t << a.log_nl) + m0;
  const uint32_t arity = 1u << a.arity_bits;
  uint64_t s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = 0;
  for (uint32_t j = 0; j < arity; j += 4) {  // 4 ext elements = 8 words per absorb
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t jj = gl::bitrev(j + u, a.arity_bits);
      const uint64_t p = base + ((uint64_t)jj << log_q);
      s[2 * u] = a.values[2 * p];
      s[2 * u + 1] = a.values[2 * p + 1];
    }
    poseidon::permute(s);
  }
  const uint64_t leaf = gl::bitrev((uint32_t)(t + ((uint64_t)m0 << a.rate_bits)), log_q + a.rate_bits);
#pragma unroll
  for (int k = 0; k < 4; k++) a.digests[leaf * 4 + k] = gl::canon(s[k]);
}
// Quad-cooperative form of the same leaf hash: lane q of a quad carries state words q and q+4, i.e.
// component (q & 1) of ext elements (q >> 1) and 2 + (q >> 1) of every 4-element absorb.
__global__ void __launch_bounds__(256) fri_layer_leaf_quad_kernel(bpg::BatchOf<bpg::FriLayerArgs> batch) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  const bpg::FriLayerArgs& a = batch.a[blockIdx.z];
  __shared__ uint64_t rc[360];
  for (uint32_t i = threadIdx.x; i < 360; i += blockDim.x) rc[i] = poseidon::RC[i];
  __syncthreads();
  const uint32_t log_q = a.log_nl - a.arity_bits;
  const uint64_t n_leaves = (uint64_t)1 << (log_q + a.rate_bits);
  const uint64_t id = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 2;
  if (id >= n_leaves) return;
  const poseidon::QuadCtx qc = poseidon::quad_ctx();
  const uint32_t t = (uint32_t)(id >> log_q), m0 = (uint32_t)(id & ((1u << log_q) - 1));
  const uint64_t base = ((uint64_t)t << a.log_nl) + m0;
  const uint32_t arity = 1u << a.arity_bits, u = qc.q >> 1, comp = qc.q & 1;
  uint64_t e[3] = {0, 0, 0};
  for (uint32_t j = 0; j < arity; j += 4) {
    const uint64_t p0 = base + ((uint64_t)gl::bitrev(j + u, a.arity_bits) << log_q);
    const uint64_t p1 = base + ((uint64_t)gl::bitrev(j + 2 + u, a.arity_bits) << log_q);
    e[0] = a.values[2 * p0 + comp];
    e[1] = a.values[2 * p1 + comp];
    poseidon::permute_quad(e, qc, rc);
  }
  const uint64_t leaf = gl::bitrev((uint32_t)(t + ((uint64_t)m0 << a.rate_bits)), log_q + a.rate_bits);
  a.digests[leaf * 4 + qc.q] = gl::canon(e[0]);
}
// Matrix-core form (poseidon_mx.cuh): a wave takes 16 * NS leaves; lane (n, kb) of set m carries state words kb and
// kb + 4 of leaf 16m + n, i.e. component (kb & 1) of ext elements (kb >> 1) and 2 + (kb >> 1) of every absorb.
template <int NS>
__global__ void __launch_bounds__(256) fri_layer_leaf_mx_kernel(bpg::BatchOf<bpg::FriLayerArgs> batch) {
  if (gridDim.x * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  const bpg::FriLayerArgs& a = batch.a[blockIdx.z];
  __shared__ __attribute__((aligned(16))) uint32_t cin[poseidon::mx::CIN_WORDS];
  poseidon::mx::build_cin(cin);
  __syncthreads();
  const poseidon::mx::Ctx c = poseidon::mx::make_ctx(cin);
  const uint32_t log_q = a.log_nl - a.arity_bits;
  const uint64_t n_leaves = (uint64_t)1 << (log_q + a.rate_bits);
  const uint64_t id0 = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (16 * NS) + (threadIdx.x & 15);
  const uint32_t arity = 1u << a.arity_bits, u = c.kb >> 1, comp = c.kb & 1;
  uint64_t base[NS], e[NS][3];
#pragma unroll
  for (int m = 0; m < NS; m++) {
    const uint64_t id = id0 + 16 * m < n_leaves ? id0 + 16 * m : n_leaves - 1;
    const uint32_t t = (uint32_t)(id >> log_q), m0 = (uint32_t)(id & ((1u << log_q) - 1));
    base[m] = ((uint64_t)t << a.log_nl) + m0;
    e[m][0] = e[m][1] = e[m][2] = 0;
  }
  for (uint32_t j = 0; j < arity; j += 4) {
    const uint64_t o0 = (uint64_t)gl::bitrev(j + u, a.arity_bits) << log_q;
    const uint64_t o1 = (uint64_t)gl::bitrev(j + 2 + u, a.arity_bits) << log_q;
#pragma unroll
    for (int m = 0; m < NS; m++) {
      e[m][0] = a.values[2 * (base[m] + o0) + comp];
      e[m][1] = a.values[2 * (base[m] + o1) + comp];
    }
    poseidon::mx::permute<NS>(e, c);
  }
#pragma unroll
  for (int m = 0; m < NS; m++) {
    const uint64_t id = id0 + 16 * m;
    if (id < n_leaves) {
      const uint32_t t = (uint32_t)(id >> log_q), m0 = (uint32_t)(id & ((1u << log_q) - 1));
      const uint64_t leaf = gl::bitrev((uint32_t)(t + ((uint64_t)m0 << a.rate_bits)), log_q + a.rate_bits);
      a.digests[leaf * 4 + c.kb] = gl::canon(e[m][0]);
    }
  }
}
// P'(x0^a) = sum_i (beta/x0)^i u_i,  u_i = 1/a * sum_j' w_a^(-i j') P(x0 w_a^j')
__global__ void __launch_bounds__(256) fri_fold_kernel(bpg::BatchOf<bpg::FriLayerArgs> batch) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  const bpg::FriLayerArgs& a = batch.a[blockIdx.z];
  const uint32_t log_q = a.log_nl - a.arity_bits;
  const uint64_t n_out = (uint64_t)1 << (log_q + a.rate_bits);
  const uint64_t id = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (id >= n_out) return;
  const uint32_t t = (uint32_t)(id >> log_q), m0 = (uint32_t)(id & ((1u << log_q) - 1));
  const uint64_t base = ((uint64_t)t << a.log_nl) + m0;
  // 1/x0 = g_t^-1 * w_{n_l}^(-m0)
  const uint64_t x_inv = gl::mulc(a.g_t_inv[t], root_pow(a.tw_nl_inv, a.log_nl, m0));
  // u_i = sum_j w_a^(-i j) v_j by four radix-2 decimation-in-frequency stages (arity = 16 is the only
  // built size, prover.cpp); u lands in bit-reversed slots.  The 1/a factor is applied once at the end.
  uint64_t c0[16], c1[16];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const uint64_t p = base + ((uint64_t)j << log_q);
    c0[j] = a.values[2 * p];
    c1[j] = a.values[2 * p + 1];
  }
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int half = 8 >> s;
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += 2) {  // two butterflies = four base-field multiplies
      uint64_t d[4], w[4], r[4];
      int mm[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int k = k0 + i, m = (k / half) * 2 * half + (k % half);
        mm[i] = m;
        const uint64_t wv = a.wa_inv_pow[(m & (half - 1)) << s];
        const uint64_t x0 = c0[m], x1 = c1[m], y0 = c0[m + half], y1 = c1[m + half];  // canonical
        c0[m] = gl::addc(x0, y0);
        c1[m] = gl::addc(x1, y1);
        d[2 * i] = gl::subc(x0, y0);
        d[2 * i + 1] = gl::subc(x1, y1);
        w[2 * i] = wv;
        w[2 * i + 1] = wv;
      }
      if (half == 1) {  // last stage: every twiddle is 1
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = d[i];
      } else {
        gl::mul_n<4>(d, w, r);
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = gl::canon(r[i]);
        // keep the groups from being interleaved: each holds ~9 live carry masks, and SGPR pressure makes the
        // compiler park masks in VGPR lanes with v_writelane right after the asm that wrote them -- a read the
        // hazard recogniser does not space out (it cannot see the write inside inline asm)
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 2; i++) {
        c0[mm[i] + half] = r[2 * i];
        c1[mm[i] + half] = r[2 * i + 1];
      }
    }
  }
  // P'(x0^a) = 1/a * sum_i (beta/x0)^i u_i, Horner from the top; u_i sits in slot bitrev4(i)
  const Ext bx = gl::scale(a.beta, x_inv);
  Ext acc = gl::ext(0);
#pragma unroll
  for (int i = 15; i >= 0; i--) {
    const int slot = ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3);
    acc = gl::add(gl::mul(acc, bx), Ext{c0[slot], c1[slot]});
  }
  acc = gl::scale(acc, a.arity_inv);
  a.out[2 * id] = acc.c0;
  a.out[2 * id + 1] = acc.c1;
}

// ---------------------------------------------------------------- K9 proof of work
// Smallest witness w >= base with leading pow_bits zero in state[7] after the duplex
// (fri_proof_of_work; upstream takes any winner, we take the minimum so results are reproducible).
__global__ void __launch_bounds__(256) pow_grind_kernel(bpg::BatchOf<bpg::PowArgs> batch, unsigned long long* result) {
  const bpg::PowArgs& a = batch.a[blockIdx.z];
  result += blockIdx.z;
  if (__hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.base) return;  // see pow_grind_mx_kernel
  const uint64_t cand = a.base + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = a.state[k];
  s[a.pos] = cand;
  poseidon::permute(s);
  if ((gl::canon(s[7]) >> (64 - a.bits)) == 0) atomicMin(result, (unsigned long long)cand);
}
// The same search with the MDS layer on the matrix cores (poseidon_mx.cuh): a wave takes 64 consecutive candidates
// as four sets of 16, lane (n, kb) holds words kb, kb + 4, kb + 8 of candidate 16m + n; word 7 is slot 1 of lanes kb = 3.
// GR: 0 = every round by itself; 2 / 3 = the partial rounds in groups (gtab = the device image of the operand tables,
// as in hash_kernels.hip).
template <int GR>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
pow_grind_mx_kernel(bpg::BatchOf<bpg::PowArgs> batch, unsigned long long* result, const uint32_t* __restrict__ gtab) {
  const bpg::PowArgs& a = batch.a[blockIdx.z];
  result += blockIdx.z;  // one witness word per proof of the batch
  // A witness below this batch is already known (an earlier batch of the same speculative group found it: batches
  // run one after the other on the stream, so the value is stable and the same for every thread): nothing here can
  // be smaller, the whole grid leaves before it loads a table.
  if (__hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.base) return;
  constexpr int NG = GR ? GR : 2;
  __shared__ __attribute__((aligned(16))) uint32_t cin[GR ? poseidon::mx::CIN_GROUPED_WORDS<NG> : poseidon::mx::CIN_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t gt[GR ? poseidon::mx::grp::TABLE_WORDS<NG> : 4];
  if constexpr (GR != 0) {
    poseidon::mx::build_cin_grouped<NG>(cin);
    poseidon::mx::grp::load_tables<NG>(gt, gtab);
  } else {
    poseidon::mx::build_cin(cin);
  }
  __syncthreads();
  const poseidon::mx::Ctx c = poseidon::mx::make_ctx(cin);
  const uint64_t cand0 = a.base + ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 15);
  uint64_t e[4][3];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    uint64_t w = a.state[4 * s];
#pragma unroll
    for (int q = 1; q < 4; q++) w = c.kb == (uint32_t)q ? a.state[4 * s + q] : w;
    const bool mine = c.kb + 4 * s == a.pos;
#pragma unroll
    for (int m = 0; m < 4; m++) e[m][s] = mine ? cand0 + 16 * m : w;
  }
  if constexpr (GR != 0) poseidon::mx::permute_grouped<NG>(e, c, gt, gtab);
  else poseidon::mx::permute<4>(e, c);
  if (c.kb == 3) {
#pragma unroll
    for (int m = 0; m < 4; m++)
      if ((gl::canon(e[m][1]) >> (64 - a.bits)) == 0) atomicMin(result, (unsigned long long)(cand0 + 16 * m));
  }
}

// ---------------------------------------------------------------- query openings
// grid = (num_queries x proofs, n_oracles).  Writes row values and the Merkle path of leaf x into the
// query record (layout: DESIGN.md section 6).
__global__ void __launch_bounds__(256) query_initial_kernel(bpg::QueryArgs a) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  const uint32_t b = blockIdx.x / a.n_queries, q = blockIdx.x % a.n_queries, o = blockIdx.y;
  const uint64_t x = a.x_index[blockIdx.x];
  const bpg::QueryOracle& orc = a.proof[b].oracle[o];
  const uint32_t log_rows = a.log_n + a.rate_bits;
  // leaf x = row bitrev: natural i = bitrev(x) = t + 2^r*m  ->  coset-major t*n + m
  const uint32_t i = gl::bitrev((uint32_t)x, log_rows);
  const uint64_t pos = ((uint64_t)(i & ((1u << a.rate_bits) - 1)) << a.log_n) + (i >> a.rate_bits);
  uint64_t* out = a.proof[b].out;
  if (o == 0 && threadIdx.x == 0) out[(uint64_t)q * a.query_words] = x;
  uint64_t* w = out + (uint64_t)q * a.query_words + orc.out_offset;
  for (uint32_t c = threadIdx.x; c < orc.n_cols; c += blockDim.x) w[c] = orc.lde[(uint64_t)c * orc.stride + pos];
  w += orc.n_cols;
  const uint32_t depth = log_rows - a.cap_height;
  for (uint32_t k = threadIdx.x; k < depth * 4; k += blockDim.x) {
    const uint32_t lvl = k >> 2;
    // level offset in digests: sum_{l<lvl} 2^(log_rows-l) = 2^(log_rows+1) - 2^(log_rows-lvl+1)
    const uint64_t off = ((uint64_t)2 << log_rows) - ((uint64_t)2 << (log_rows - lvl));
    w[k] = orc.digests[(off + ((x >> lvl) ^ 1)) * 4 + (k & 3)];
  }
  // what a parent circuit walks from (proofgen.cpp: first_query_trace_path): the digest of the leaf itself, which the
  // proof does not carry and the host would otherwise re-hash from the opened row (304 permutations for 2432 columns)
  if (q == 0 && o == a.leaf_oracle && a.proof[b].first_leaf && threadIdx.x < 4) a.proof[b].first_leaf[threadIdx.x] = orc.digests[x * 4 + threadIdx.x];
}
// grid = (num_queries x proofs, n_layers)
__global__ void __launch_bounds__(64) query_layers_kernel(bpg::QueryLayerArgs a) {
  if (gridDim.x * gridDim.y * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  const uint32_t b = blockIdx.x / a.n_queries, q = blockIdx.x % a.n_queries, l = blockIdx.y;
  const bpg::QueryLayer& L = a.proof[b].layer[l];
  const uint32_t arity = 1u << a.arity_bits;
  const uint64_t leaf = a.x_index[blockIdx.x] >> (a.arity_bits * (l + 1));
  const uint32_t log_q = L.log_nl - a.arity_bits, log_leaves = log_q + a.rate_bits;
  const uint32_t c = gl::bitrev((uint32_t)leaf, log_leaves);  // = t + 2^r * m0
  const uint32_t t = c & ((1u << a.rate_bits) - 1), m0 = c >> a.rate_bits;
  const uint64_t base = ((uint64_t)t << L.log_nl) + m0;
  uint64_t* w = a.proof[b].out + (uint64_t)q * a.query_words + L.out_offset;
  for (uint32_t k = threadIdx.x; k < 2 * arity; k += blockDim.x) {
    const uint32_t jj = gl::bitrev(k >> 1, a.arity_bits);
    w[k] = L.values[2 * (base + ((uint64_t)jj << log_q)) + (k & 1)];
  }
  w += 2 * arity;
  const uint32_t depth = log_leaves - a.cap_height;
  for (uint32_t k = threadIdx.x; k < depth * 4; k += blockDim.x) {
    const uint32_t lvl = k >> 2;
    const uint64_t off = ((uint64_t)2 << log_leaves) - ((uint64_t)2 << (log_leaves - lvl));
    w[k] = L.digests[(off + ((leaf >> lvl) ^ 1)) * 4 + (k & 3)];
  }
}

}  // namespace

// ================================================================= host-side launchers
namespace bpg {

int launch_synth_constants(uint64_t* d_out, uint32_t log_n, uint32_t n_const, uint64_t seed, hipStream_t st) {
  uint64_t total = (uint64_t)n_const << log_n;
  if (!total) return BP_OK;
  synth_const_kernel<<<ceil_div(total, 256), 256, 0, st>>>(d_out, log_n, n_const, seed);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
static int check_batch(uint32_t batch) {
  return batch >= 1 && batch <= MAX_BATCH ? BP_OK : fail(BP_ERR_INVALID_INPUT, "batch of %u proofs: 1..%u can be proved in lock-step", batch, MAX_BATCH);
}
int launch_synth_trace(const SynthTraceArgs* a, uint32_t batch, uint32_t log_n, uint32_t n_cols, uint32_t n_const,
                       uint32_t deg_pow, hipStream_t st) {
  if (int rc = check_batch(batch)) return rc;
  dim3 grid(ceil_div((uint64_t)1 << log_n, 256), n_cols / 4 + 1, batch);
  synth_trace_kernel<<<grid, 256, 0, st>>>(batch_of(a, batch), log_n, n_cols, n_const, deg_pow);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_keccak_trace(uint64_t* d_trace, const uint64_t* d_inputs, uint32_t log_n, uint64_t seed, hipStream_t st) {
  dim3 grid(ceil_div((uint64_t)1 << log_n, 256), 6);
  keccak_trace_kernel<<<grid, 256, 0, st>>>(d_trace, d_inputs, log_n, seed);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_logic_trace(uint64_t* d_trace, const uint64_t* d_inputs, uint32_t log_n, uint64_t seed, hipStream_t st) {
  dim3 grid(ceil_div((uint64_t)1 << log_n, 256), 2);
  logic_trace_kernel<<<grid, 256, 0, st>>>(d_trace, d_inputs, log_n, seed);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_memory_trace(uint64_t* d_trace, const uint64_t* d_inputs, uint32_t log_n, uint64_t seed, hipStream_t st) {
  memory_trace_kernel<<<ceil_div((uint64_t)1 << log_n, 256), 256, 0, st>>>(d_trace, d_inputs, log_n, seed);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_arithmetic_trace(uint64_t* d_trace, const uint64_t* d_inputs, uint32_t log_n, uint64_t seed, hipStream_t st) {
  arithmetic_trace_kernel<<<ceil_div((uint64_t)1 << log_n, 256), 256, 0, st>>>(d_trace, d_inputs, log_n, seed);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_byte_packing_trace(uint64_t* d_trace, const uint64_t* d_inputs, uint32_t log_n, uint64_t seed, hipStream_t st) {
  byte_packing_trace_kernel<<<ceil_div((uint64_t)1 << log_n, 256), 256, 0, st>>>(d_trace, d_inputs, log_n, seed);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_keccak_sponge_trace(uint64_t* d_trace, const uint64_t* d_inputs, uint32_t log_n, uint64_t seed, hipStream_t st,
                               uint32_t row_limit) {
  dim3 grid(ceil_div((uint64_t)1 << log_n, 256), 3);
  keccak_sponge_trace_kernel<<<grid, 256, 0, st>>>(d_trace, d_inputs, log_n, seed, row_limit);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_keccak_inputs_from_sponge(const uint64_t* d_sponge_trace, uint32_t sponge_log_n, uint64_t* d_inputs, uint32_t n_perms,
                                     uint64_t seed, hipStream_t st) {
  if (!n_perms) return BP_OK;
  keccak_inputs_from_sponge_kernel<<<ceil_div(n_perms, 256), 256, 0, st>>>(d_sponge_trace, sponge_log_n, d_inputs, n_perms, seed);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_logic_inputs_from_sponge(const uint64_t* d_sponge_trace, uint32_t sponge_log_n, uint32_t covered, const uint64_t* d_given,
                                    uint32_t n_given, uint64_t* d_inputs, uint32_t n_logic, uint64_t seed, hipStream_t st) {
  if (5ull * covered > n_logic || covered > (1u << sponge_log_n))
    return fail(BP_ERR_INVALID_INPUT, "the logic table (%u rows) cannot hold five operations for each of %u sponge rows", n_logic, covered);
  logic_inputs_from_sponge_kernel<<<ceil_div(n_logic, 256), 256, 0, st>>>(d_sponge_trace, sponge_log_n, covered, d_given, n_given, d_inputs,
                                                                           n_logic, seed);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_memory_inputs_from_byte_packing(const uint64_t* d_pack_trace, uint32_t pack_log_n, uint64_t* d_log, uint32_t n_mem,
                                           hipStream_t st) {
  memory_inputs_from_byte_packing_kernel<<<ceil_div(n_mem, 256), 256, 0, st>>>(d_pack_trace, pack_log_n, d_log, n_mem);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_lookup_filter(uint32_t air_id, uint64_t* d_trace, uint32_t log_n, const uint64_t* flag_a, const uint64_t* flag_b,
                         uint32_t n_flags, hipStream_t st) {
  if (!flag_a || !n_flags) return BP_OK;  // nothing is exposed: the trace kernels left the column zero
  if (air_id == air::KECCAK_F) {
    if (!flag_b) return fail(BP_ERR_INVALID_INPUT, "launch_lookup_filter: the Keccak-f table's filter needs both flag columns of the sponge table");
    keccak_lookup_filter_kernel<<<ceil_div((uint64_t)1 << log_n, 256), 256, 0, st>>>(d_trace, log_n, flag_a, flag_b, n_flags);
  } else if (air_id == air::MEMORY) {
    memory_lookup_filter_kernel<<<ceil_div(n_flags, 256), 256, 0, st>>>(d_trace, log_n, flag_a, n_flags);
  } else if (air_id == air::LOGIC) {  // flag_a / flag_b: the sponge table's is_full / is_final columns, n_flags: covered sponge rows
    if (!flag_b) return fail(BP_ERR_INVALID_INPUT, "launch_lookup_filter: the logic table's filter needs both flag columns of the sponge table");
    logic_lookup_filter_kernel<<<ceil_div(5ull * n_flags, 256), 256, 0, st>>>(d_trace, log_n, flag_a, flag_b, n_flags);
  } else {
    return fail(BP_ERR_INVALID_INPUT, "launch_lookup_filter: AIR %u is not a looked table", air_id);
  }
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_arithmetic_mul_trace(uint64_t* d_trace, const uint64_t* d_inputs, uint32_t log_n, uint64_t seed, hipStream_t st) {
  arithmetic_mul_trace_kernel<<<ceil_div((uint64_t)1 << log_n, 256), 256, 0, st>>>(d_trace, d_inputs, log_n, seed);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_plonk_constants(uint64_t* d_out, uint32_t log_n, uint64_t seed, const air::plonk::Layout& lay, hipStream_t st) {
  if (lay.pi_len < 1 || lay.pi_len > air::plonk::MAX_PI)
    return fail(BP_ERR_INVALID_INPUT, "a recursion circuit hashes a public-input list of 1..%u words: got %u", air::plonk::MAX_PI, lay.pi_len);
  if (log_n >= 4 && log_n < 31 && !air::plonk::layout_ok(lay, 1u << log_n))
    return fail(BP_ERR_INVALID_INPUT, "plonk circuit layout: %u paths of %u levels (at most %u Merkle rows), their words at %u.. of a "
                "list of %u, leaves of %u words (0 or more than 8; at most %u leaf rows), and one arithmetic group must fit 2^%u rows",
                lay.n_paths, lay.depth, air::plonk::MERKLE_ROWS_MAX, lay.path_pi0, lay.pi_len, lay.leaf_len, air::plonk::LEAF_ROWS_MAX, log_n);
  if (log_n < 5) return fail(BP_ERR_INVALID_INPUT, "the plonk circuit needs 32 rows: the Poseidon rows (4..19) and one arithmetic group");
  const uint64_t* tw_n = nullptr;
  if (int rc = get_table(0, log_n, 0, &tw_n)) return rc;
  plonk_constants_kernel<<<dim3(ceil_div((uint64_t)1 << log_n, 256), air::plonk::N_CONST), 256, 0, st>>>(d_out, log_n, seed, tw_n, lay);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_plonk_trace(const PlonkTraceArgs* a, uint32_t batch, uint32_t log_n, hipStream_t st) {
  if (int rc = check_batch(batch)) return rc;
  for (uint32_t b = 0; b < batch; b++)
    if (!a[b].hash_rows || a[b].n_hash_rows < 1 || a[b].n_hash_rows > air::plonk::HASH_ROWS_MAX)
      return fail(BP_ERR_INVALID_INPUT, "launch_plonk_trace: the witness of the hash rows is missing (poseidon_hash_rows)");
  uint32_t region = 0;
  for (uint32_t b = 0; b < batch; b++) {
    if (a[b].arith_row0 < air::plonk::MERKLE_ROW0 || (a[b].arith_row0 & 3) || a[b].arith_row0 + 4 > (1ull << log_n) ||
        air::plonk::MERKLE_ROW0 + a[b].n_merkle_rows > a[b].arith_row0 || a[b].n_merkle_rows > air::plonk::MERKLE_ROWS_MAX + air::plonk::LEAF_ROWS_MAX)
      return fail(BP_ERR_INVALID_INPUT, "launch_plonk_trace: %u Merkle rows and first arithmetic row %u do not fit 2^%u rows", a[b].n_merkle_rows, a[b].arith_row0, log_n);
    region = std::max(region, a[b].arith_row0 - air::plonk::HASH_ROW0);
  }
  plonk_trace_kernel<<<dim3(ceil_div((((uint64_t)1 << log_n) / 4) * air::plonk::N_SLOTS, 256), 1, batch), 256, 0, st>>>(batch_of(a, batch), log_n);
  BPG_LAUNCH_CHECK();
  plonk_hash_rows_kernel<<<dim3(1, region, batch), 256, 0, st>>>(batch_of(a, batch), log_n);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_aux(const AuxArgs* a, uint32_t batch, uint32_t air_id, uint32_t n_cols, uint32_t log_n, hipStream_t st) {
  const air::Shape shape{air_id, n_cols, 0, 1};
  const uint32_t n_aux = air::ctl::n_aux(shape), p0 = air::ctl::first_product(air_id);
  if (!n_aux) return BP_OK;
  if (int rc = check_batch(batch)) return rc;
  const BatchOf<AuxArgs> ab = batch_of(a, batch);
  if (air_id == air::KECCAK_F) {  // the helper columns first: the products read them
    keccak_ctl_helpers_kernel<<<dim3(ceil_div((uint64_t)1 << log_n, 256), 1, batch), 256, 0, st>>>(ab, log_n);
    BPG_LAUNCH_CHECK();
  }
  uint32_t threads = (1u << log_n) < 1024 ? (1u << log_n) : 1024;
  if (threads < 64) threads = 64;
  if (air_id == air::PLONK) {  // chunk ratios per row, suffix products of the row totals, partial products
    for (uint32_t b = 0; b < batch; b++)
      if (!a[b].consts) return fail(BP_ERR_INVALID_INPUT, "the plonk AIR needs the circuit's constant columns (the sigmas) for its copy products");
    const uint64_t* tw_n = nullptr;
    if (int rc = get_table(0, log_n, 0, &tw_n)) return rc;
    const uint32_t gx = ceil_div((uint64_t)1 << log_n, 256);
    plonk_chunk_ratios_kernel<<<dim3(gx, 2, batch), 256, 0, st>>>(ab, log_n, tw_n);
    BPG_LAUNCH_CHECK();
    aux_suffix_product_kernel<air::PLONK><<<dim3(2, 1, batch), threads, 0, st>>>(ab, log_n, n_cols);
    BPG_LAUNCH_CHECK();
    plonk_partial_products_kernel<<<dim3(gx, 18, batch), 256, 0, st>>>(ab, log_n);
    BPG_LAUNCH_CHECK();
    return BP_OK;
  }
  const dim3 grid(n_aux - p0, 1, batch);
  // algorithmic bytes: every column a product reads, once, and the product column written (SURVEY.md section 8(d):
  // 24 n per column of a synthetic table)
  const double reads = air_id == air::SYNTHETIC ? 2 : air_id == air::KECCAK_F ? 53 : air_id == air::KECCAK_SPONGE ? 102
                       : air_id == air::BYTE_PACKING ? 43 : air_id == air::MEMORY ? 12 : air_id == air::LOGIC ? 524 : 0;
  KernelTimer kt(PROF_AUX, st, 8.0 * (double)((uint64_t)1 << log_n) * (reads + 1) * (n_aux - p0) * batch, true);
  switch (air_id) {
    case air::SYNTHETIC: BPG_LAUNCH_TIMED(kt, aux_suffix_product_kernel<air::SYNTHETIC>, grid, threads, 0, st, ab, log_n, n_cols); break;
    case air::KECCAK_F: BPG_LAUNCH_TIMED(kt, aux_suffix_product_kernel<air::KECCAK_F>, grid, threads, 0, st, ab, log_n, n_cols); break;
    case air::KECCAK_SPONGE: BPG_LAUNCH_TIMED(kt, aux_suffix_product_kernel<air::KECCAK_SPONGE>, grid, threads, 0, st, ab, log_n, n_cols); break;
    case air::BYTE_PACKING: BPG_LAUNCH_TIMED(kt, aux_suffix_product_kernel<air::BYTE_PACKING>, grid, threads, 0, st, ab, log_n, n_cols); break;
    case air::MEMORY: BPG_LAUNCH_TIMED(kt, aux_suffix_product_kernel<air::MEMORY>, grid, threads, 0, st, ab, log_n, n_cols); break;
    case air::LOGIC: BPG_LAUNCH_TIMED(kt, aux_suffix_product_kernel<air::LOGIC>, grid, threads, 0, st, ab, log_n, n_cols); break;
    default: BPG_LAUNCH_TIMED(kt, aux_suffix_product_kernel<air::ARITHMETIC>, grid, threads, 0, st, ab, log_n, n_cols); break;  // no lookup: z = 1
  }
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_quotient(const QuotArgs* qs, uint32_t batch, const QuotCoset& coset, hipStream_t st) {
  if (int rc = check_batch(batch)) return rc;
  const QuotArgs& q = qs[0];
  const uint64_t rows = (uint64_t)1 << (q.log_n + q.rate_bits);
  // the alpha-power table of every proof's two challenges (a few thousand words, read wave-uniformly)
  CosetTab ct;
  for (int t = 0; t < 16; t++) { ct.v[t] = coset.g_t[t]; ct.v[16 + t] = coset.zh_t[t]; ct.v[32 + t] = coset.zh_inv_t[t]; }
  BatchOf<AlphaTabArgs> at{};
  for (uint32_t b = 0; b < batch; b++) at.a[b] = AlphaTabArgs{const_cast<uint64_t*>(qs[b].apow), qs[b].alpha0, qs[b].alpha1};
  alpha_table_kernel<<<dim3(ceil_div(q.n_constraints, 256), 2, batch), 256, 0, st>>>(at, q.n_constraints, ct);
  BPG_LAUNCH_CHECK();
  const uint32_t n_units = q.n_air_units + q.n_ctl_units, wg_rows = ceil_div(n_units, q.units_per_wg);
  const BatchOf<QuotArgs> qb = batch_of(qs, batch);
  dim3 g1(ceil_div(rows, 256), wg_rows, batch);
  // algorithmic bytes: every element of the three LDE matrices read once, the two quotient columns written
  // (AIR 8 is counted with the synthetic recursion-shaped proofs it replaces)
  KernelTimer kt(PROF_K5 + (q.air_id < air::COUNT ? q.air_id : 0), st, 8.0 * (double)rows * ((double)q.n_cols + q.n_aux + q.n_const + 2) * batch, true);
  if (q.air_id == bpg::air::KECCAK_F) BPG_LAUNCH_TIMED(kt, quotient_air_kernel<bpg::air::KECCAK_F>, g1, 256, 0, st, qb);
  else if (q.air_id == bpg::air::LOGIC) BPG_LAUNCH_TIMED(kt, quotient_air_kernel<bpg::air::LOGIC>, g1, 256, 0, st, qb);
  else if (q.air_id == bpg::air::MEMORY) BPG_LAUNCH_TIMED(kt, quotient_air_kernel<bpg::air::MEMORY>, g1, 256, 0, st, qb);
  else if (q.air_id == bpg::air::ARITHMETIC) BPG_LAUNCH_TIMED(kt, quotient_air_kernel<bpg::air::ARITHMETIC>, g1, 256, 0, st, qb);
  else if (q.air_id == bpg::air::BYTE_PACKING) BPG_LAUNCH_TIMED(kt, quotient_air_kernel<bpg::air::BYTE_PACKING>, g1, 256, 0, st, qb);
  else if (q.air_id == bpg::air::KECCAK_SPONGE) BPG_LAUNCH_TIMED(kt, quotient_air_kernel<bpg::air::KECCAK_SPONGE>, g1, 256, 0, st, qb);
  else if (q.air_id == bpg::air::ARITHMETIC_MUL) BPG_LAUNCH_TIMED(kt, quotient_air_kernel<bpg::air::ARITHMETIC_MUL>, g1, 256, 0, st, qb);
  else if (q.air_id == bpg::air::PLONK) BPG_LAUNCH_TIMED(kt, quotient_air_kernel<bpg::air::PLONK>, g1, 256, 0, st, qb);
  else BPG_LAUNCH_TIMED(kt, quotient_air_kernel<bpg::air::SYNTHETIC>, g1, 256, 0, st, qb);
  kt.stop();
  if (q.side_rows) {
    BPG_LAUNCH_CHECK();
    // same family, no algorithmic bytes of its own: the wires it reads are the ones the chunk units have just read
    KernelTimer kh(PROF_K5 + q.air_id, st, 0.0, true);
    BPG_LAUNCH_TIMED(kh, quotient_plonk_hash_kernel, dim3(ceil_div(rows, 256), 1, batch), 256, 0, st, qb, wg_rows);
  }
  BPG_LAUNCH_CHECK();
  if (wg_rows + q.side_rows > 1) {
    quotient_sum_kernel<<<dim3(ceil_div(rows, 256), 2, batch), 256, 0, st>>>(qb, wg_rows + q.side_rows);
    BPG_LAUNCH_CHECK();
  }
  return BP_OK;
}
int launch_quotient_chunks(const ChunkArgs* c, uint32_t batch, hipStream_t st) {
  if (int rc = check_batch(batch)) return rc;
  dim3 grid(ceil_div((uint64_t)1 << c[0].log_n, 256), 2, batch);
  quotient_chunks_kernel<<<grid, 256, 0, st>>>(batch_of(c, batch));
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_power_vectors(const PowerVecArgs* a, uint32_t batch, uint32_t log_n, uint32_t n_points, hipStream_t st) {
  if (int rc = check_batch(batch)) return rc;
  if (n_points < 1 || n_points > 3) return fail(BP_ERR_INVALID_INPUT, "launch_power_vectors: 1..3 points");
  dim3 grid(ceil_div((uint64_t)1 << log_n, 256), n_points, batch);
  power_vector_kernel<<<grid, 256, 0, st>>>(batch_of(a, batch), log_n);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_alpha_pows(const AlphaPowArgs* a, uint32_t batch, uint32_t count, hipStream_t st) {
  if (int rc = check_batch(batch)) return rc;
  alpha_pows_kernel<<<dim3(ceil_div(count, 256), 1, batch), 256, 0, st>>>(batch_of(a, batch), count);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_openings(const uint64_t* d_coeffs, uint64_t stride, uint32_t log_n, uint32_t n_cols,
                    const uint64_t* d_pw, uint32_t n_points, uint64_t* d_out, hipStream_t st) {
  if (!n_cols) return BP_OK;
  openings_kernel<<<n_cols, 256, 0, st>>>(d_coeffs, stride, log_n, d_pw, n_points, d_out);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_openings_multi(const OpenMulti* m, uint32_t batch, hipStream_t st) {
  if (int rc = check_batch(batch)) return rc;
  const uint32_t total = m[0].first_col[m[0].n_segs];  // the proofs of a batch have one shape
  if (!total) return BP_OK;
  KernelTimer kt(PROF_OPENINGS, st, 8.0 * (double)((uint64_t)1 << m[0].log_n) * total * batch, true);  // every coefficient column once
  BPG_LAUNCH_TIMED(kt, openings_multi_kernel, dim3(total, 1, batch), 256, 0, st, batch_of(m, batch));
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
// algorithmic bytes of the alpha-combination of one proof: every coefficient column read once, six result columns written
static double combine_bytes(const CombineMulti& m) {
  double cols = 6;
  for (uint32_t o = 0; o < m.n_oracles; o++) cols += m.a[o].n_cols;
  return 8.0 * (double)((uint64_t)1 << m.a[0].log_n) * cols;
}
int launch_combine_partial_multi(const CombineMulti* m, uint32_t batch, uint32_t total_chunks, hipStream_t st) {
  if (!total_chunks) return BP_OK;
  if (int rc = check_batch(batch)) return rc;
  dim3 grid(ceil_div((uint64_t)1 << m[0].a[0].log_n, 256), total_chunks, batch);
  KernelTimer kt(PROF_FRI_COMBINE, st, combine_bytes(m[0]) * batch, true);
  BPG_LAUNCH_TIMED(kt, fri_combine_partial_multi_kernel, grid, 256, 0, st, batch_of(m, batch));
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_combine_all(const CombineMulti* m, uint32_t batch, uint64_t* const* d_g, hipStream_t st) {
  if (int rc = check_batch(batch)) return rc;
  GPtrs gp{};
  for (uint32_t b = 0; b < batch; b++) gp.g[b] = d_g[b];
  KernelTimer kt(PROF_FRI_COMBINE, st, combine_bytes(m[0]) * batch, true);
  BPG_LAUNCH_TIMED(kt, fri_combine_all_kernel, dim3(ceil_div((uint64_t)1 << m[0].a[0].log_n, 256), 1, batch), 256, 0, st, batch_of(m, batch), gp);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_combine_reduce(const CombineReduceArgs* a, uint32_t batch, uint32_t n_chunks, uint32_t log_n, hipStream_t st) {
  if (int rc = check_batch(batch)) return rc;
  dim3 grid(ceil_div((uint64_t)1 << log_n, 256), 6, batch);
  fri_combine_reduce_kernel<<<grid, 256, 0, st>>>(batch_of(a, batch), n_chunks, log_n);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_fri_init(const FriInitArgs* a, uint32_t batch, hipStream_t st) {
  if (int rc = check_batch(batch)) return rc;
  uint64_t rows = (uint64_t)1 << (a[0].log_n + a[0].rate_bits);
  fri_quotient_values_kernel<<<dim3(ceil_div(rows, 256), 1, batch), 256, 0, st>>>(batch_of(a, batch));
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
uint64_t quad_threshold();  // hash_kernels.hip
bool poseidon_mx();         // hash_kernels.hip
int mx_sets(uint64_t n);    // hash_kernels.hip
int launch_fri_layer_leaves(const FriLayerArgs* as, uint32_t batch, hipStream_t st) {
  if (int rc = check_batch(batch)) return rc;
  const FriLayerArgs& a = as[0];
  uint64_t n = (uint64_t)1 << (a.log_nl - a.arity_bits + a.rate_bits);
  const BatchOf<FriLayerArgs> ab = batch_of(as, batch);
  // the form follows the work of the whole launch (every proof has its own grid.z slice: no wave spans two proofs)
  if (const int ns = mx_sets(n * batch)) {
    if (ns == 4) fri_layer_leaf_mx_kernel<4><<<dim3(ceil_div(n, 256), 1, batch), 256, 0, st>>>(ab);
    else if (ns == 2) fri_layer_leaf_mx_kernel<2><<<dim3(ceil_div(n, 128), 1, batch), 256, 0, st>>>(ab);
    else fri_layer_leaf_mx_kernel<1><<<dim3(ceil_div(n, 64), 1, batch), 256, 0, st>>>(ab);
  } else if (n * batch < quad_threshold()) fri_layer_leaf_quad_kernel<<<dim3(ceil_div(n * 4, 256), 1, batch), 256, 0, st>>>(ab);
  else fri_layer_leaf_kernel<<<dim3(ceil_div(n, 256), 1, batch), 256, 0, st>>>(ab);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_fri_fold(const FriLayerArgs* a, uint32_t batch, hipStream_t st) {
  if (int rc = check_batch(batch)) return rc;
  uint64_t n = (uint64_t)1 << (a[0].log_nl - a[0].arity_bits + a[0].rate_bits);
  // 16 M (1 + 1/arity): the layer's M extension values read, M / arity written
  KernelTimer kt(PROF_FRI_FOLD, st, 16.0 * (double)n * ((1 << a[0].arity_bits) + 1) * batch, true);
  BPG_LAUNCH_TIMED(kt, fri_fold_kernel, dim3(ceil_div(n, 256), 1, batch), 256, 0, st, batch_of(a, batch));
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_pow(const PowArgs* a, uint32_t batch, uint32_t n_candidates, unsigned long long* d_result, hipStream_t st) {
  if (int rc = check_batch(batch)) return rc;
  const BatchOf<PowArgs> ab = batch_of(a, batch);
  const dim3 grid(n_candidates / 256, 1, batch);
  if (poseidon_mx()) {
    int ng = 0;
    const uint32_t* gtab = group_tables(&ng);
    if (gtab && ng == 3) pow_grind_mx_kernel<3><<<grid, 256, 0, st>>>(ab, d_result, gtab);
    else if (gtab) pow_grind_mx_kernel<2><<<grid, 256, 0, st>>>(ab, d_result, gtab);
    else pow_grind_mx_kernel<0><<<grid, 256, 0, st>>>(ab, d_result, nullptr);
  } else
    pow_grind_kernel<<<grid, 256, 0, st>>>(ab, d_result);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_query_initial(const QueryArgs& a, uint32_t batch, uint32_t n_oracles, hipStream_t st) {
  if (int rc = check_batch(batch)) return rc;
  if (!a.n_queries || a.n_queries * batch > MAX_BATCH_QUERIES) return fail(BP_ERR_INVALID_INPUT, "query launch: %u x %u indices, at most %u", a.n_queries, batch, MAX_BATCH_QUERIES);
  dim3 grid(a.n_queries * batch, n_oracles);
  query_initial_kernel<<<grid, 256, 0, st>>>(a);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
int launch_query_layers(const QueryLayerArgs& a, uint32_t batch, uint32_t n_layers, hipStream_t st) {
  if (!n_layers) return BP_OK;
  if (int rc = check_batch(batch)) return rc;
  if (!a.n_queries || a.n_queries * batch > MAX_BATCH_QUERIES || n_layers > MAX_FRI_LAYERS)
    return fail(BP_ERR_INVALID_INPUT, "query launch: %u x %u indices, %u layers", a.n_queries, batch, n_layers);
  dim3 grid(a.n_queries * batch, n_layers);
  query_layers_kernel<<<grid, 64, 0, st>>>(a);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}

}  // namespace bpg
