// hash_kernels.hip -- K3/K4: Poseidon permutation batch, Merkle leaf hashing and tree build.
//
// Restates plonky2::hash::{poseidon, hashing::hash_n_to_m_no_pad, merkle_tree::MerkleTree::new}
// as reached from plonky_block_proof_gen/src/proof_gen.rs:44-52 (PolynomialBatch::from_values).
// Leaf hashing streams the LDE matrix once (8*L*C bytes, coalesced 8-byte column loads: lane =
// row, so a wave reads 512 contiguous bytes per column) but is integer-ALU bound: ceil(C/8)
// permutations per row.
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>
#include "common.hpp"
#include "poseidon.cuh"
#include "poseidon_mx.cuh"
#include "stark_kernels.hpp"

namespace {

__global__ void __launch_bounds__(256) perm_batch_kernel(uint64_t* __restrict__ states, uint64_t n) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = states[i * 12 + k];
  poseidon::permute(s);
#pragma unroll
  for (int k = 0; k < 12; k++) states[i * 12 + k] = gl::canon(s[k]);
}

// One lane per LDE row.  Row `pos` = t*n + m (coset-major) is Merkle leaf bitrev_r(t)*n +
// bitrev_n(m) (upstream reverse_index_bits order); the 32-byte digest is scattered there.
// >= 5 waves per SIMD: the interleaved S-boxes otherwise balloon to ~190 VGPRs (2 waves), which exposes latency
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
leaf_hash_kernel(const uint64_t* __restrict__ lde, uint64_t stride, uint32_t n_cols, uint32_t log_n,
                 uint32_t rate_bits, uint64_t* __restrict__ digests, uint64_t lde_bstride, uint64_t dig_bstride) {
  lde += blockIdx.z * lde_bstride;  // tree blockIdx.z of a batch of equally shaped commitments
  digests += blockIdx.z * dig_bstride;
  const uint64_t rows = (uint64_t)1 << (log_n + rate_bits);
  uint64_t pos = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (pos >= rows) return;
  const uint32_t t = (uint32_t)(pos >> log_n), m = (uint32_t)(pos & ((1u << log_n) - 1));
  const uint64_t leaf = ((uint64_t)gl::bitrev(t, rate_bits) << log_n) | gl::bitrev(m, log_n);
  uint64_t s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = 0;
  const uint64_t* p = lde + pos;
  if (n_cols <= 4) {  // hash_or_noop: short rows are the digest
    for (uint32_t c = 0; c < n_cols; c++) s[c] = p[(uint64_t)c * stride];
  } else {
    uint32_t c = 0;
    for (; c + 8 <= n_cols; c += 8) {
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] = p[(uint64_t)(c + k) * stride];
      poseidon::permute(s);
    }
    if (c < n_cols) {  // ragged tail: overwrite only the first (n_cols - c) rate words
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (c + k < n_cols) s[k] = p[(uint64_t)(c + k) * stride];
      poseidon::permute(s);
    }
  }
  uint64_t* d = digests + leaf * 4;
#pragma unroll
  for (int k = 0; k < 4; k++) d[k] = gl::canon(s[k]);
}

// Quad-cooperative variants (poseidon.cuh): four lanes per row / node.  blockDim = 256 = 64 quads.
__global__ void __launch_bounds__(256)
leaf_hash_quad_kernel(const uint64_t* __restrict__ lde, uint64_t stride, uint32_t n_cols, uint32_t log_n,
                      uint32_t rate_bits, uint64_t* __restrict__ digests, uint64_t lde_bstride, uint64_t dig_bstride) {
  lde += blockIdx.z * lde_bstride;
  digests += blockIdx.z * dig_bstride;
  __shared__ uint64_t rc[360];
  for (uint32_t i = threadIdx.x; i < 360; i += blockDim.x) rc[i] = poseidon::RC[i];
  __syncthreads();
  const uint64_t rows = (uint64_t)1 << (log_n + rate_bits);
  const uint64_t pos = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 2;
  if (pos >= rows) return;  // whole quads leave together (rows is a power of two >= 1, 64 quads per block)
  const poseidon::QuadCtx qc = poseidon::quad_ctx();
  const uint32_t q = qc.q;
  const uint32_t t = (uint32_t)(pos >> log_n), m = (uint32_t)(pos & ((1u << log_n) - 1));
  const uint64_t leaf = ((uint64_t)gl::bitrev(t, rate_bits) << log_n) | gl::bitrev(m, log_n);
  const uint64_t* p = lde + pos;
  uint64_t e[3] = {0, 0, 0};
  if (n_cols <= 4) {
    if (q < n_cols) e[0] = p[(uint64_t)q * stride];
  } else {
    for (uint32_t c = 0; c < n_cols; c += 8) {
      if (c + q < n_cols) e[0] = p[(uint64_t)(c + q) * stride];
      if (c + 4 + q < n_cols) e[1] = p[(uint64_t)(c + 4 + q) * stride];
      poseidon::permute_quad(e, qc, rc);
    }
  }
  digests[leaf * 4 + q] = gl::canon(e[0]);
}
__global__ void __launch_bounds__(256)
merkle_level_quad_kernel(const uint64_t* __restrict__ child, uint64_t* __restrict__ parent, uint64_t n_parents,
                         uint64_t* __restrict__ mirror, uint64_t dig_bstride) {
  if (gridDim.x * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  child += blockIdx.z * dig_bstride;
  parent += blockIdx.z * dig_bstride;
  if (mirror) mirror += blockIdx.z * n_parents * 4;  // the mirrored level is the cap: 4 * n_parents words per tree
  __shared__ uint64_t rc[360];
  for (uint32_t i = threadIdx.x; i < 360; i += blockDim.x) rc[i] = poseidon::RC[i];
  __syncthreads();
  const uint64_t i = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 2;
  if (i >= n_parents) return;
  const poseidon::QuadCtx qc = poseidon::quad_ctx();
  uint64_t e[3] = {child[i * 8 + qc.q], child[i * 8 + 4 + qc.q], 0};
  poseidon::permute_quad(e, qc, rc);
  const uint64_t d = gl::canon(e[0]);
  parent[i * 4 + qc.q] = d;
  if (mirror) mirror[i * 4 + qc.q] = d;  // cap level: also straight into the host-visible mailbox
}

// Several Merkle levels in one launch: each 256-lane workgroup (64 quads) owns 64 consecutive parents
// of the first level and everything above them, handing digests down through LDS.  Every level is
// still written to the level-order digest buffer (Merkle paths need all of them).  Replaces up to
// seven single-level launches on a proof's critical path.
__global__ void __launch_bounds__(256)
merkle_subtree_quad_kernel(const uint64_t* __restrict__ child, uint64_t* __restrict__ out, uint64_t n_parents,
                           uint32_t levels, uint64_t* __restrict__ mirror, uint64_t dig_bstride) {
  __shared__ uint64_t rc[360];
  __shared__ uint64_t sm[2][64 * 4];
  if (gridDim.x * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);
  child += blockIdx.z * dig_bstride;
  out += blockIdx.z * dig_bstride;
  if (mirror) mirror += blockIdx.z * ((n_parents >> (levels - 1)) * 4);  // the last level of the launch is the cap
  for (uint32_t i = threadIdx.x; i < 360; i += blockDim.x) rc[i] = poseidon::RC[i];
  __syncthreads();
  const poseidon::QuadCtx qc = poseidon::quad_ctx();
  const uint32_t quad = threadIdx.x >> 2;
  uint64_t level_parents = n_parents;                    // parents of the level being produced (whole tree)
  uint32_t wg_parents = n_parents < 64 ? (uint32_t)n_parents : 64u;  // ... and in this workgroup
  uint64_t wg_first = (uint64_t)blockIdx.x * 64;         // index of this workgroup's first parent in that level
  uint64_t* dst = out;
  for (uint32_t l = 0; l < levels; l++) {
    if (quad < wg_parents) {
      uint64_t e[3];
      if (l == 0) {
        const uint64_t p = wg_first + quad;
        e[0] = child[p * 8 + qc.q];
        e[1] = child[p * 8 + 4 + qc.q];
      } else {
        e[0] = sm[(l - 1) & 1][(2 * quad) * 4 + qc.q];
        e[1] = sm[(l - 1) & 1][(2 * quad + 1) * 4 + qc.q];
      }
      e[2] = 0;
      poseidon::permute_quad(e, qc, rc);
      const uint64_t d = gl::canon(e[0]);
      dst[(wg_first + quad) * 4 + qc.q] = d;
      if (mirror && l + 1 == levels) mirror[(wg_first + quad) * 4 + qc.q] = d;
      sm[l & 1][quad * 4 + qc.q] = d;
    }
    __syncthreads();
    dst += level_parents * 4;
    level_parents >>= 1;
    wg_parents >>= 1;
    wg_first >>= 1;
  }
}

// Row-major leaves (FRI layers): leaf k = leaf_len consecutive words.
// >= 5 waves per SIMD: the interleaved S-boxes otherwise balloon to ~190 VGPRs (2 waves), which exposes latency
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
leaf_hash_rows_kernel(const uint64_t* __restrict__ leaves, uint32_t leaf_len, uint64_t n_leaves,
                      uint64_t* __restrict__ digests) {
  if (gridDim.x * gridDim.y <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (k >= n_leaves) return;
  const uint64_t* p = leaves + k * leaf_len;
  uint64_t s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = 0;
  if (leaf_len <= 4) {
    for (uint32_t c = 0; c < leaf_len; c++) s[c] = p[c];
  } else {
    for (uint32_t c = 0; c < leaf_len; c += 8) {
#pragma unroll
      for (int i = 0; i < 8; i++)
        if (c + i < leaf_len) s[i] = p[c + i];
      poseidon::permute(s);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) digests[k * 4 + i] = gl::canon(s[i]);
}

// One lane per parent node of one level.
__global__ void __launch_bounds__(256)
merkle_level_kernel(const uint64_t* __restrict__ child, uint64_t* __restrict__ parent, uint64_t n_parents,
                    uint64_t* __restrict__ mirror, uint64_t dig_bstride) {
  if (gridDim.x * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  child += blockIdx.z * dig_bstride;
  parent += blockIdx.z * dig_bstride;
  if (mirror) mirror += blockIdx.z * n_parents * 4;
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= n_parents) return;
  uint64_t l[4], r[4], o[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    l[k] = child[i * 8 + k];
    r[k] = child[i * 8 + 4 + k];
  }
  poseidon::two_to_one(l, r, o);
#pragma unroll
  for (int k = 0; k < 4; k++) parent[i * 4 + k] = o[k];
  if (mirror) {
#pragma unroll
    for (int k = 0; k < 4; k++) mirror[i * 4 + k] = o[k];
  }
}

// ---- matrix-core ("mx") form, poseidon_mx.cuh: a wave owns 16 * NS rows / nodes / states as NS sets of 16; lane
// (n = lane & 15, kb = lane >> 4) holds words kb, kb + 4, kb + 8 of item 16m + n of every set m.  No lane leaves before
// the permutations are done (an MFMA is a whole-wave instruction): out-of-range items are clamped for the loads and
// masked at the store.  NS = 4 is the throughput form, NS = 1 (4x the waves, like the quad form, at 0.65x its
// instructions) the one for launches that cannot fill the chip.
// At least three waves per SIMD (<= 168 VGPRs): with the grouped rounds' operands the four-set kernels otherwise take
// 172..192 VGPRs, two waves per SIMD, and lose more to the exposed MFMA / LDS latency than the groups save.
// Larger workgroups do not pay: four waves per SIMD (512 threads, 128 VGPRs, ~25 registers in scratch) hash 2.92
// instead of 2.98 Gperm/s and gain nothing under load (profiles/r3_poseidon_occ4_ab.txt); six- and twelve-wave
// workgroups (to share a 67 KB table image) lose a quarter of their resident waves (r3_poseidon_three_groups.txt).
#define BPG_MX_BOUNDS __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
// GR (NS = 4 only): 0 = every round by itself; 2 = the partial rounds 4..19 in two groups of eight; 3 = all 22 partial
// rounds in groups (8 + 8 + 6).  gtab = the device image of the operand tables (poseidon_mx.cuh, grp), copied into LDS.
#define BPG_MX_TABLES(NS, GR, gtab)                                                                           \
  static_assert(!(GR) || (NS) == 4, "groups exist for four sets per wave");                                   \
  __shared__ __attribute__((aligned(16)))                                                                     \
      uint32_t cin[(GR) ? poseidon::mx::CIN_GROUPED_WORDS<(GR) ? (GR) : 2> : poseidon::mx::CIN_WORDS];        \
  __shared__ __attribute__((aligned(16))) uint32_t gt[(GR) ? poseidon::mx::grp::TABLE_WORDS<(GR) ? (GR) : 2> : 4]; \
  if constexpr ((GR) != 0) {                                                                                  \
    poseidon::mx::build_cin_grouped<(GR) ? (GR) : 2>(cin);                                                    \
    poseidon::mx::grp::load_tables<(GR) ? (GR) : 2>(gt, gtab);                                                \
  } else {                                                                                                    \
    poseidon::mx::build_cin(cin);                                                                             \
  }                                                                                                           \
  __syncthreads();
#define BPG_MX_PERMUTE(NS, GR, e, c, gtab)                                                      \
  if constexpr ((GR) != 0) poseidon::mx::permute_grouped<(GR) ? (GR) : 2>(e, c, gt, gtab);  \
  else poseidon::mx::permute<NS>(e, c);

template <int NS, int GR>
__global__ void BPG_MX_BOUNDS perm_batch_mx_kernel(uint64_t* __restrict__ states, uint64_t n,
                                                            const uint32_t* __restrict__ gtab) {
  BPG_MX_TABLES(NS, GR, gtab)
  const poseidon::mx::Ctx c = poseidon::mx::make_ctx(cin);
  const uint64_t base = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (16 * NS) + (threadIdx.x & 15);
  uint64_t e[NS][3];
#pragma unroll
  for (int m = 0; m < NS; m++) {
    const uint64_t i = base + 16 * m < n ? base + 16 * m : n - 1;
#pragma unroll
    for (int a = 0; a < 3; a++) e[m][a] = states[i * 12 + c.kb + 4 * a];
  }
  BPG_MX_PERMUTE(NS, GR, e, c, gtab)
#pragma unroll
  for (int m = 0; m < NS; m++) {
    const uint64_t i = base + 16 * m;
    if (i < n) {
#pragma unroll
      for (int a = 0; a < 3; a++) states[i * 12 + c.kb + 4 * a] = gl::canon(e[m][a]);
    }
  }
}

template <int NS, int GR>
__global__ void BPG_MX_BOUNDS
leaf_hash_mx_kernel(const uint64_t* __restrict__ lde, uint64_t stride, uint32_t n_cols, uint32_t log_n,
                    uint32_t rate_bits, uint64_t* __restrict__ digests, uint64_t lde_bstride, uint64_t dig_bstride,
                    const uint32_t* __restrict__ gtab) {
  if (gridDim.x * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  lde += blockIdx.z * lde_bstride;
  digests += blockIdx.z * dig_bstride;
  BPG_MX_TABLES(NS, GR, gtab)
  const poseidon::mx::Ctx c = poseidon::mx::make_ctx(cin);
  const uint64_t rows = (uint64_t)1 << (log_n + rate_bits);
  const uint64_t base = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (16 * NS) + (threadIdx.x & 15);
  const uint32_t kb = c.kb;
  // rows is a power of two >= 16 * NS here (the launcher picks NS by size), so the sets of a wave are 16 rows apart
  // and none is out of range unless the whole wave is: one pointer, immediate offsets
  const uint64_t* p0 = lde + (base < rows ? base : (uint64_t)(threadIdx.x & 15));
  const uint64_t* p[NS];
  uint64_t e[NS][3];
#pragma unroll
  for (int m = 0; m < NS; m++) {
    p[m] = p0 + 16 * m;
    e[m][0] = e[m][1] = e[m][2] = 0;
  }
  if (n_cols <= 4) {  // hash_or_noop: short rows are the digest
    if (kb < n_cols) {
#pragma unroll
      for (int m = 0; m < NS; m++) e[m][0] = p[m][(uint64_t)kb * stride];
    }
  } else {
    for (uint32_t col = 0; col < n_cols; col += 8) {  // wave-uniform trip count
      if (col + kb < n_cols) {
#pragma unroll
        for (int m = 0; m < NS; m++) e[m][0] = p[m][(uint64_t)(col + kb) * stride];
      }
      if (col + 4 + kb < n_cols) {
#pragma unroll
        for (int m = 0; m < NS; m++) e[m][1] = p[m][(uint64_t)(col + 4 + kb) * stride];
      }
      BPG_MX_PERMUTE(NS, GR, e, c, gtab)
    }
  }
#pragma unroll
  for (int m = 0; m < NS; m++) {
    const uint64_t pos = base + 16 * m;
    if (pos < rows) {
      const uint32_t t = (uint32_t)(pos >> log_n), mm = (uint32_t)(pos & ((1u << log_n) - 1));
      const uint64_t leaf = ((uint64_t)gl::bitrev(t, rate_bits) << log_n) | gl::bitrev(mm, log_n);
      digests[leaf * 4 + kb] = gl::canon(e[m][0]);
    }
  }
}

template <int NS, int GR>
__global__ void BPG_MX_BOUNDS
merkle_level_mx_kernel(const uint64_t* __restrict__ child, uint64_t* __restrict__ parent, uint64_t n_parents,
                       uint64_t* __restrict__ mirror, uint64_t dig_bstride, const uint32_t* __restrict__ gtab) {
  if (gridDim.x * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);  // small launch = latency-critical: issue first
  child += blockIdx.z * dig_bstride;
  parent += blockIdx.z * dig_bstride;
  if (mirror) mirror += blockIdx.z * n_parents * 4;
  BPG_MX_TABLES(NS, GR, gtab)
  const poseidon::mx::Ctx c = poseidon::mx::make_ctx(cin);
  const uint64_t base = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (16 * NS) + (threadIdx.x & 15);
  uint64_t e[NS][3];
#pragma unroll
  for (int m = 0; m < NS; m++) {
    const uint64_t i = base + 16 * m < n_parents ? base + 16 * m : n_parents - 1;
    e[m][0] = child[i * 8 + c.kb];
    e[m][1] = child[i * 8 + 4 + c.kb];
    e[m][2] = 0;
  }
  BPG_MX_PERMUTE(NS, GR, e, c, gtab)
#pragma unroll
  for (int m = 0; m < NS; m++) {
    const uint64_t i = base + 16 * m;
    if (i < n_parents) {
      const uint64_t d = gl::canon(e[m][0]);
      parent[i * 4 + c.kb] = d;
      if (mirror) mirror[i * 4 + c.kb] = d;
    }
  }
}

// Up to seven Merkle levels in one launch, matrix-core form with ONE set of 16 states per wave (the low-latency form:
// ~3.8 k VALU instructions per permutation and wave instead of 12.6 k).  A workgroup of four waves takes 64 parents
// of the first level from global memory and hands every level down through LDS: 64 -> 32 -> ... -> 1 (or until the
// cap); every level is still written to the level-order digest buffer (Merkle paths need all of them).  Replaces
// up to seven latency-bound single-level launches on a proof's critical path (merkle_subtree_quad_kernel is the same
// with the quad-cooperative permutation, ~12 us per level against ~8).
__global__ void __launch_bounds__(256)
merkle_subtree_mx_kernel(const uint64_t* __restrict__ child, uint64_t* __restrict__ out, uint64_t n_parents,
                         uint32_t levels, uint64_t* __restrict__ mirror, uint64_t dig_bstride) {
  if (gridDim.x * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);
  child += blockIdx.z * dig_bstride;
  out += blockIdx.z * dig_bstride;
  if (mirror) mirror += blockIdx.z * ((n_parents >> (levels - 1)) * 4);  // the last level of the launch is the cap
  __shared__ __attribute__((aligned(16))) uint32_t cin[poseidon::mx::CIN_WORDS];
  __shared__ uint64_t sm[2][64 * 4];
  poseidon::mx::build_cin(cin);
  __syncthreads();
  const poseidon::mx::Ctx c = poseidon::mx::make_ctx(cin);
  const uint32_t wave = threadIdx.x >> 6, n = threadIdx.x & 15;
  uint64_t level_parents = n_parents;                                  // parents of the level being produced
  uint32_t wg_parents = n_parents < 64 ? (uint32_t)n_parents : 64u;    // ... and in this workgroup
  uint64_t wg_first = (uint64_t)blockIdx.x * 64;
  uint64_t* dst = out;
  for (uint32_t l = 0; l < levels; l++) {
    if (wave * 16 < wg_parents) {  // whole waves only: an MFMA is a wave-wide instruction
      const uint32_t q = wave * 16 + n, qc = q < wg_parents ? q : wg_parents - 1;
      uint64_t e[1][3];
      if (l == 0) {
        e[0][0] = child[(wg_first + qc) * 8 + c.kb];
        e[0][1] = child[(wg_first + qc) * 8 + 4 + c.kb];
      } else {
        e[0][0] = sm[(l - 1) & 1][(2 * qc) * 4 + c.kb];
        e[0][1] = sm[(l - 1) & 1][(2 * qc + 1) * 4 + c.kb];
      }
      e[0][2] = 0;
      poseidon::mx::permute<1>(e, c);
      if (q < wg_parents) {
        const uint64_t d = gl::canon(e[0][0]);
        dst[(wg_first + q) * 4 + c.kb] = d;
        if (mirror && l + 1 == levels) mirror[(wg_first + q) * 4 + c.kb] = d;
        sm[l & 1][q * 4 + c.kb] = d;
      }
    }
    __syncthreads();
    dst += level_parents * 4;
    level_parents >>= 1;
    wg_parents >>= 1;
    wg_first >>= 1;
  }
}

// The same hand-down for the levels ABOVE that: a workgroup takes 256 parents of its first level and reduces them nine
// levels (256 -> 1, or until the cap).  While a level has 64 or more parents in the workgroup its waves carry four
// sets of 16 states each (the throughput form: 256 / 128 / 64 parents = 4 / 2 / 1 waves at work), below that one set
// per wave as in merkle_subtree_mx_kernel.  A 2^16-leaf tree is then three launches -- the widest level by itself,
// this kernel from 2^14 nodes, the cap's last levels -- instead of six.
__global__ void __launch_bounds__(256)
merkle_subtree_wide_kernel(const uint64_t* __restrict__ child, uint64_t* __restrict__ out, uint64_t n_parents,
                           uint32_t levels, uint64_t* __restrict__ mirror, uint64_t dig_bstride) {
  if (gridDim.x * gridDim.z <= 64) __builtin_amdgcn_s_setprio(3);
  child += blockIdx.z * dig_bstride;
  out += blockIdx.z * dig_bstride;
  if (mirror) mirror += blockIdx.z * ((n_parents >> (levels - 1)) * 4);
  __shared__ __attribute__((aligned(16))) uint32_t cin[poseidon::mx::CIN_WORDS];
  __shared__ uint64_t sm[2][256 * 4];
  poseidon::mx::build_cin(cin);
  __syncthreads();
  const poseidon::mx::Ctx c = poseidon::mx::make_ctx(cin);
  const uint32_t wave = threadIdx.x >> 6, n = threadIdx.x & 15;
  uint64_t level_parents = n_parents;
  uint32_t wg_parents = n_parents < 256 ? (uint32_t)n_parents : 256u;
  uint64_t wg_first = (uint64_t)blockIdx.x * 256;
  uint64_t* dst = out;
  for (uint32_t l = 0; l < levels; l++) {
    const uint64_t* prev = sm[(l - 1) & 1];
    if (wg_parents >= 64) {
      if (wave * 64 < wg_parents) {  // whole waves only
        uint64_t e[4][3];
#pragma unroll
        for (int m = 0; m < 4; m++) {
          const uint32_t q = wave * 64 + 16 * m + n;  // < wg_parents: it is a multiple of 64 here
          if (l == 0) {
            e[m][0] = child[(wg_first + q) * 8 + c.kb];
            e[m][1] = child[(wg_first + q) * 8 + 4 + c.kb];
          } else {
            e[m][0] = prev[(2 * q) * 4 + c.kb];
            e[m][1] = prev[(2 * q + 1) * 4 + c.kb];
          }
          e[m][2] = 0;
        }
        poseidon::mx::permute<4>(e, c);
#pragma unroll
        for (int m = 0; m < 4; m++) {
          const uint32_t q = wave * 64 + 16 * m + n;
          const uint64_t d = gl::canon(e[m][0]);
          dst[(wg_first + q) * 4 + c.kb] = d;
          if (mirror && l + 1 == levels) mirror[(wg_first + q) * 4 + c.kb] = d;
          sm[l & 1][q * 4 + c.kb] = d;
        }
      }
    } else if (wave * 16 < wg_parents) {
      const uint32_t q = wave * 16 + n, qc = q < wg_parents ? q : wg_parents - 1;
      uint64_t e[1][3];
      if (l == 0) {
        e[0][0] = child[(wg_first + qc) * 8 + c.kb];
        e[0][1] = child[(wg_first + qc) * 8 + 4 + c.kb];
      } else {
        e[0][0] = prev[(2 * qc) * 4 + c.kb];
        e[0][1] = prev[(2 * qc + 1) * 4 + c.kb];
      }
      e[0][2] = 0;
      poseidon::mx::permute<1>(e, c);
      if (q < wg_parents) {
        const uint64_t d = gl::canon(e[0][0]);
        dst[(wg_first + q) * 4 + c.kb] = d;
        if (mirror && l + 1 == levels) mirror[(wg_first + q) * 4 + c.kb] = d;
        sm[l & 1][q * 4 + c.kb] = d;
      }
    }
    __syncthreads();
    dst += level_parents * 4;
    level_parents >>= 1;
    wg_parents >>= 1;
    wg_first >>= 1;
  }
}

// 8-byte-per-lane streaming copy: calibrates the rocprofv3 FETCH_SIZE / WRITE_SIZE counters for the
// access width every field kernel here uses (MI355X_MICROARCH.md, HBM section).
__global__ void __launch_bounds__(256)
calib_copy_u64_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t n) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}

// K1 check kernel: every device form of the field arithmetic on caller-supplied operand pairs.
// out[op][i] (15 planes); a is ANY u64 (also non-canonical), b likewise for the multiplies; the add/sub forms get
// canon(b) as the contract of gl::add / gl::sub demands.
__global__ void __launch_bounds__(256) field_ops_kernel(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b,
                                                        uint64_t* __restrict__ out, uint64_t n) {
  const uint64_t i0 = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 4;
  if (i0 >= n) return;
  uint64_t x[4], y[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint64_t i = i0 + k < n ? i0 + k : n - 1;
    x[k] = a[i];
    y[k] = b[i];
  }
  uint64_t r4[4], r3[3];
  gl::mul_n<4>(x, y, r4);
  __builtin_amdgcn_sched_barrier(0);
  const uint64_t x3[3] = {x[1], x[2], x[3]}, y3[3] = {y[1], y[2], y[3]};
  gl::mul_n<3>(x3, y3, r3);
  __builtin_amdgcn_sched_barrier(0);  // one group of carry masks at a time (SGPR pressure, see stark_kernels.hip fri_fold)
  gl::DotAcc d[4] = {gl::dot_zero(), gl::dot_zero(), gl::dot_zero(), gl::dot_zero()};
  gl::dot_mad4(d, x, y);
  __builtin_amdgcn_sched_barrier(0);
  gl::dot_mad4(d, y, x);           // 2*a*b, through the wrap counters when the operands are large
  __builtin_amdgcn_sched_barrier(0);
  const uint64_t yy[4] = {y[0], y[0], y[0], y[0]};
  gl::dot_mad4(d, x, yy);          // + a_k * b_0
  __builtin_amdgcn_sched_barrier(0);
  // the interleaved group forms of the lazy add / sub / canonicalisation (the NTT butterflies' arithmetic)
  uint64_t cy[4] = {y[0], y[1], y[2], y[3]}, sum4[4], dif4[4], cx4[4] = {x[0], x[1], x[2], x[3]};
  gl::canon_n<4>(cy);
  __builtin_amdgcn_sched_barrier(0);
  gl::add_n<4>(x, cy, sum4);
  __builtin_amdgcn_sched_barrier(0);
  gl::sub_n<4>(x, cy, dif4);
  __builtin_amdgcn_sched_barrier(0);
  gl::canon_n<4>(cx4);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint64_t i = i0 + k;
    if (i >= n) break;
    const uint64_t cb = gl::canon(y[k]);
    out[0 * n + i] = gl::canon(gl::mul(x[k], y[k]));                 // one-element carry chain
    out[1 * n + i] = gl::canon(r4[k]);                               // groups of four
    out[2 * n + i] = k ? gl::canon(r3[k - 1]) : gl::canon(r4[0]);    // groups of three
    uint64_t lo, hi;
    gl::mul_wide(x[k], y[k], lo, hi);
    out[3 * n + i] = gl::canon(gl::reduce128(lo, hi));               // compiler form
    out[4 * n + i] = gl::canon(gl::add(x[k], cb));
    out[5 * n + i] = gl::canon(gl::sub(x[k], cb));
    out[6 * n + i] = gl::dot_reduce(d[k]);                           // 2*a*b + a*b_0 of the group
    out[7 * n + i] = gl::canon(gl::mul7(x[k]));
    out[8 * n + i] = gl::inv(gl::canon(x[k]));
    const gl::Ext e = gl::mul(gl::Ext{gl::canon(x[k]), cb}, gl::Ext{cb, gl::canon(x[k] ^ y[k])});
    out[9 * n + i] = e.c0;
    out[10 * n + i] = e.c1;
    out[11 * n + i] = gl::canon(sum4[k]);   // add_n<4>: a + canon(b), lazily reduced
    out[12 * n + i] = gl::canon(dif4[k]);   // sub_n<4>
    out[13 * n + i] = cx4[k];               // canon_n<4>(a): must already be canonical
    out[14 * n + i] = cy[k];
  }
}

}  // namespace

namespace bpg {


// launches with fewer permutations than this use the quad-cooperative kernels (4x the waves)
// 0 = automatic: the quad form (4x the waves, 1.22x the instructions) pays while the chip is not full, so
// the threshold follows the number of provers at work: few -> 2^17 (measured alone: quad wins up to there),
// many -> 2^13 (under 24-stream load the instruction count decides; 2^11..2^13 measured best by ~1 %).
static std::atomic<uint64_t> g_quad_threshold{0};
static bool g_poseidon_mx_on();
static std::atomic<int> g_active_provers{0};
void prover_active(int delta) { g_active_provers.fetch_add(delta, std::memory_order_relaxed); }
int provers_active() { return g_active_provers.load(std::memory_order_relaxed); }
static std::atomic<int> g_assume_loaded{-1};  // -1: by the count of provers at work; 0 / 1: stated by the caller
bool device_loaded() {  // several provers share the chip
  const int a = g_assume_loaded.load(std::memory_order_relaxed);
  return a < 0 ? g_active_provers.load(std::memory_order_relaxed) >= 6 : a != 0;
}
uint64_t quad_threshold() {
  const uint64_t t = g_quad_threshold.load(std::memory_order_relaxed);
  if (t) return t;
  const bool loaded = device_loaded();
  // with the matrix-core forms the small-launch form (one set per wave) costs 0.65x the quad form's instructions
  // and wins alone up to 2^18 items (tools/kernel_bench.py: 2^17 rows 1.8-2.0 against 1.5-1.8 Gperm/s for four sets)
  if (g_poseidon_mx_on()) return (uint64_t)1 << (loaded ? 13 : 19);
  return (uint64_t)1 << (loaded ? 13 : 17);
}
// Levels near the root are each one latency-bound launch (a lone txn proof spends ~30 % of its kernel time in them,
// and under the 24-stream load they are 40 % of all launches, each stretched from 19 to ~120 us by sharing:
// profiles/r2b_kernel_stats_4txn_1stream.csv, r3_kernel_stats_64txn_24streams.csv).  merkle_subtree_mx_kernel hands
// up to seven levels of at most 2048 nodes down through LDS in one launch, in the one-set matrix-core form.
// Measured in round 3 (profiles/r3_small_shards.txt), fused against one launch per level: 256 txns 36.2 against 35.2
// txn-proofs/s, 32 txns 33.6 against 32.5, 16 txns 33.1 against 31.6, a lone pair of txns 150.8 against 148.2 ms.
// (With the quad-cooperative permutation -- round 2's fused kernel, still used when the matrix-core forms are switched
// off -- the fused form lost 0-4 %: ~12 us per level cost what the launch gaps saved.)
// 1 = fused (default), 0 = one launch per level, -1 = fused only while fewer than six provers are at work.
static std::atomic<int> g_merkle_fused{1};
static std::atomic<int> g_merkle_wide_log2{0};  // levels of up to 2^k parents go to merkle_subtree_wide_kernel (0: none)
// launches at or above the quad threshold: 1 = matrix-core form (poseidon_mx.cuh), 0 = one lane per state
static std::atomic<int> g_poseidon_mx{1};
bool poseidon_mx() { return g_poseidon_mx.load(std::memory_order_relaxed) != 0; }
static bool g_poseidon_mx_on() { return poseidon_mx(); }
// Sets of 16 states per wave for an mx launch of n items (0 = matrix-core form off).  Four sets are the throughput
// form (all 64 lanes busy in the partial-round S-box); a launch below the quad threshold takes fewer sets = more
// waves, as the quad form did (which the one-set form replaces at 0.65x the instructions).
static std::atomic<int> g_mx_sets{0};  // 0 = by size, else 1 / 2 / 4
int mx_sets(uint64_t n) {
  if (!poseidon_mx()) return 0;
  const int f = g_mx_sets.load(std::memory_order_relaxed);
  if (f) return f;
  const uint64_t t = quad_threshold();
  return n >= t ? 4 : (n >= t / 2 ? 2 : 1);
}
// ---- operand tables of the grouped partial rounds (poseidon_group.hpp): built once per process on the host,
// uploaded once per device; the four-set kernels copy them into LDS.  nullptr = per-round form (knob, or no tables).
static std::atomic<int> g_poseidon_grouped{3};  // 0, 2 or 3 groups (bp_tune_poseidon_grouped)
static std::mutex g_group_mu;
static const uint32_t* g_group_dev[2][16] = {};
static std::atomic<int> g_group_state[2][16];  // [n_groups - 2][device]: 0 not tried, 1 ready, -1 failed
static std::vector<uint32_t>* g_group_image[2] = {nullptr, nullptr};

template <int NG>
static bool build_group_image() {
  namespace pg = poseidon::group;
  namespace gx = poseidon::mx::grp;
  if (g_group_image[NG - 2]) return true;
  auto img = std::make_unique<std::vector<uint32_t>>(gx::IMAGE_WORDS<NG>, 0);
  for (int g = 0; g < NG; g++) {
    pg::Tables t;
    uint32_t* c = img->data() + gx::OPS_WORDS + g * gx::C_WORDS;
    if (g < 2) {
      if (!pg::build(gx::K, 4 + gx::K * g, &t) || (int)t.ops.size() != gx::LAY.n_ops * 1024) return false;
      if (g == 0) std::memcpy(img->data(), t.ops.data(), t.ops.size());
      else if (std::memcmp(img->data(), t.ops.data(), t.ops.size()) != 0) return false;  // the A operands do not depend on r0
    } else {
      // the short group (rounds 20..25): the form operands are the eight-round group's -- a six-round group's are
      // the same rows with forms 6 and 7 blank, checked here --, its MAIN operands and C tables are its own
      constexpr pg::Layout L6 = pg::layout(gx::SHORT_K);
      if (!pg::build(gx::SHORT_K, 20, &t) || (int)t.ops.size() != L6.n_ops * 1024) return false;
      const uint8_t* o8 = reinterpret_cast<const uint8_t*>(img->data());
      // operand (pair 0: identical; pair 1: W and the delta of sigma_4) of the short layout -> of the long one
      for (int i = 0; i < L6.main_base; i++) {
        const int j = i < L6.d_base[1] ? i : gx::LAY.d_base[1] + (i - L6.d_base[1]);
        for (int b = 0; b < 1024; b++) {
          const uint8_t v6 = t.ops[(size_t)i * 1024 + b], v8 = o8[(size_t)j * 1024 + b];
          const int row = (b >> 4) & 15;  // lane = b >> 4, row = lane & 15: rows 8..15 of pair 1 are forms 6 and 7
          const bool blank = i >= L6.w_base[1] && row >= 8;
          if (blank ? v6 != 0 : v6 != v8) return false;
        }
      }
      std::memcpy(img->data() + gx::TABLE_WORDS<NG>, t.ops.data() + (size_t)L6.main_base * 1024, 18 * 1024);
    }
    std::memcpy(c, t.cform.data(), pg::CFORM_WORDS * 4);
    std::memcpy(c + pg::CFORM_WORDS, t.cmain.data(), pg::CMAIN_WORDS * 4);
  }
  // the per-round MDS layer's A operands as poseidon::mx::make_ctx builds them: lane (r, kb), dword a =
  // M[(r >> 2) + 4g][kb + 4a] << 8 (r & 3)
  static const uint32_t MC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  uint32_t* am = img->data() + gx::OPS_WORDS + NG * gx::C_WORDS;
  for (uint32_t g = 0; g < 3; g++)
    for (uint32_t lane = 0; lane < 64; lane++)
      for (uint32_t a = 0; a < 3; a++) {
        const uint32_t r = lane & 15, kb = lane >> 4, i = (r >> 2) + 4 * g, k = kb + 4 * a;
        am[(g * 64 + lane) * 4 + a] = (MC[(k + 12 - i) % 12] + ((i | k) == 0 ? 8u : 0u)) << (8 * (r & 3));
      }
  g_group_image[NG - 2] = img.release();
  return true;
}
// the current device's copy of the image (uploaded on first use) and the number of groups it serves, or nullptr
const uint32_t* group_tables(int* n_groups) {
  const int ng = g_poseidon_grouped.load(std::memory_order_relaxed);
  if (n_groups) *n_groups = ng;
  if (ng != 2 && ng != 3) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::atomic<int>& state = g_group_state[ng - 2][dev];
  const uint32_t*& slot = g_group_dev[ng - 2][dev];
  if (state.load(std::memory_order_acquire) > 0) return slot;  // the state is set after the pointer
  std::lock_guard<std::mutex> lk(g_group_mu);
  if (const int st = state.load(std::memory_order_acquire)) return st > 0 ? slot : nullptr;
  void* d = nullptr;
  const size_t bytes = (size_t)(ng == 3 ? poseidon::mx::grp::IMAGE_WORDS<3> : poseidon::mx::grp::IMAGE_WORDS<2>) * 4;
  if (!(ng == 3 ? build_group_image<3>() : build_group_image<2>()) || hipMalloc(&d, bytes) != hipSuccess ||
      hipMemcpy(d, g_group_image[ng - 2]->data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
    if (d) (void)hipFree(d);
    state.store(-1, std::memory_order_release);
    return nullptr;
  }
  slot = static_cast<const uint32_t*>(d);
  state.store(1, std::memory_order_release);
  return slot;
}

// ITEMS: rows / nodes / states of ONE tree; BATCH trees (grid.z) of that size in the launch
#define BPG_MX_DISPATCH(NS_EXPR, KERNEL, ITEMS, BATCH, ...)                                                 \
  switch (NS_EXPR) {                                                                                       \
    case 4: {                                                                                              \
      int ng_ = 0;                                                                                         \
      const uint32_t* gtab_ = bpg::group_tables(&ng_);                                                     \
      if (gtab_ && ng_ == 3)                                                                               \
        KERNEL<4, 3><<<dim3(ceil_div((ITEMS), 256), 1, (BATCH)), 256, 0, st>>>(__VA_ARGS__, gtab_);       \
      else if (gtab_)                                                                                      \
        KERNEL<4, 2><<<dim3(ceil_div((ITEMS), 256), 1, (BATCH)), 256, 0, st>>>(__VA_ARGS__, gtab_);       \
      else                                                                                                 \
        KERNEL<4, 0><<<dim3(ceil_div((ITEMS), 256), 1, (BATCH)), 256, 0, st>>>(__VA_ARGS__, nullptr);     \
      break;                                                                                               \
    }                                                                                                      \
    case 2: KERNEL<2, 0><<<dim3(ceil_div((ITEMS), 128), 1, (BATCH)), 256, 0, st>>>(__VA_ARGS__, nullptr); break; \
    default: KERNEL<1, 0><<<dim3(ceil_div((ITEMS), 64), 1, (BATCH)), 256, 0, st>>>(__VA_ARGS__, nullptr); break; \
  }

// `mirror` (nullable): host-visible buffer that receives the 2^cap_height cap digests directly from
// the kernel that produces them, so the caller needs no device->host copy, only a stream wait.
// Returns with *mirrored = false when there was no level to run (the cap is the leaf level).
int merkle_upper_levels(uint64_t* d_digests, uint32_t log_leaves, uint32_t cap_height, hipStream_t st,
                        uint64_t* mirror, bool* mirrored, uint32_t batch, uint64_t dig_bstride) {
  uint64_t* lvl = d_digests;
  uint32_t l = log_leaves;
  if (mirrored) *mirrored = false;
  if (batch == 0 || batch > MAX_BATCH) return fail(BP_ERR_INVALID_INPUT, "merkle: batch of %u trees", batch);
  // `batch` trees of this shape are in every launch (grid.z): the choice of kernel form goes by the work of the
  // whole launch, the fusing limits by one tree (a workgroup never spans two trees)
  while (l > cap_height) {
    const uint64_t cnt = (uint64_t)1 << l, parents = cnt / 2;
    uint64_t* nxt = lvl + cnt * 4;
    // fuse only what fits in <= 64 workgroups: those launches are latency-critical and run at raised priority
    const int fmode = g_merkle_fused.load(std::memory_order_relaxed);
    // from 2^11 nodes down (measured 2^10 / 2^11 / 2^12 / 2^13 / 2^14: 36.9 / 37.0 / 36.8 / 35.2 / 34.7 txn-proofs/s: wider
    // levels are throughput work for the four-set kernels); modes 8..20 of the knob = fuse from 2^mode nodes down
    const uint64_t flimit = fmode >= 8 ? (uint64_t)1 << fmode : 2048;
    const bool fused = parents <= flimit && (parents < quad_threshold() || fmode >= 8) && (fmode > 0 || (fmode < 0 && !device_loaded()));
    // the levels above the one-set fused tail, 256 parents per workgroup, nine levels per launch (knob: wide limit)
    const uint64_t wlimit = (uint64_t)1 << g_merkle_wide_log2.load(std::memory_order_relaxed);
    if (!fused && fmode > 0 && poseidon_mx() && parents >= 256 && parents <= wlimit && parents % 256 == 0) {
      uint32_t levels = l - cap_height;
      if (levels > 9) levels = 9;
      uint64_t* mir = (l - levels == cap_height) ? mirror : nullptr;
      merkle_subtree_wide_kernel<<<dim3((uint32_t)(parents / 256), 1, batch), 256, 0, st>>>(lvl, nxt, parents, levels, mir, dig_bstride);
      BPG_LAUNCH_CHECK();
      if (mir && mirrored) *mirrored = true;
      for (uint32_t k = 0; k < levels; k++) {
        lvl += ((uint64_t)1 << l) * 4;
        l--;
      }
      continue;
    }
    if (!fused) {
      uint64_t* mir = (l - 1 == cap_height) ? mirror : nullptr;
      if (const int ns = mx_sets(parents * batch)) {
        BPG_MX_DISPATCH(ns, merkle_level_mx_kernel, parents, batch, lvl, nxt, parents, mir, dig_bstride)
      } else if (parents * batch >= quad_threshold())  // big level: one lane per node
        merkle_level_kernel<<<dim3(ceil_div(parents, 256), 1, batch), 256, 0, st>>>(lvl, nxt, parents, mir, dig_bstride);
      else
        merkle_level_quad_kernel<<<dim3(ceil_div(cnt * 2, 256), 1, batch), 256, 0, st>>>(lvl, nxt, parents, mir, dig_bstride);
      BPG_LAUNCH_CHECK();
      if (mir && mirrored) *mirrored = true;
      lvl = nxt;
      l--;
      continue;
    }
    // fused: a workgroup's 64 parents can be reduced 7 levels (64 -> 1); fewer if the cap comes first
    // or the level is smaller than one workgroup
    uint32_t levels = l - cap_height;
    const uint32_t wgs = parents >= 64 ? (uint32_t)(parents / 64) : 1;
    uint32_t max_levels = 1;
    for (uint64_t p = parents >= 64 ? 64 : parents; p > 1; p >>= 1) max_levels++;
    if (levels > max_levels) levels = max_levels;
    uint64_t* mir = (l - levels == cap_height) ? mirror : nullptr;
    if (poseidon_mx()) merkle_subtree_mx_kernel<<<dim3(wgs, 1, batch), 256, 0, st>>>(lvl, nxt, parents, levels, mir, dig_bstride);
    else merkle_subtree_quad_kernel<<<dim3(wgs, 1, batch), 256, 0, st>>>(lvl, nxt, parents, levels, mir, dig_bstride);
    BPG_LAUNCH_CHECK();
    if (mir && mirrored) *mirrored = true;
    for (uint32_t k = 0; k < levels; k++) {
      lvl += ((uint64_t)1 << l) * 4;
      l--;
    }
  }
  return BP_OK;
}

int merkle_commit_rows(const uint64_t* d_leaves, uint32_t leaf_len, uint32_t log_leaves, uint32_t cap_height,
                       uint64_t* d_digests, hipStream_t st) {
  uint64_t n = (uint64_t)1 << log_leaves;
  leaf_hash_rows_kernel<<<ceil_div(n, 256), 256, 0, st>>>(d_leaves, leaf_len, n, d_digests);
  BPG_LAUNCH_CHECK();
  return merkle_upper_levels(d_digests, log_leaves, cap_height, st, nullptr, nullptr);
}

}  // namespace bpg

extern "C" {

int bp_debug_copy_u64(const uint64_t* d_in, uint64_t* d_out, uint64_t n, void* stream) {
  if (!d_in || !d_out) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_debug_copy_u64: null buffer");
  calib_copy_u64_kernel<<<4096, 256, 0, bpg::as_stream(stream)>>>(d_in, d_out, n);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}

int bp_debug_field_ops(const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, uint64_t n, void* stream) {
  if (!n) return BP_OK;
  if (!d_a || !d_b || !d_out) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_debug_field_ops: null buffer");
  field_ops_kernel<<<bpg::ceil_div(bpg::ceil_div(n, 4), 256), 256, 0, bpg::as_stream(stream)>>>(d_a, d_b, d_out, n);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}

void bp_tune_merkle_fused(int mode) { bpg::g_merkle_fused.store(mode < 0 ? -1 : (mode >= 8 && mode <= 20 ? mode : (mode != 0))); }
void bp_tune_quad_threshold(uint64_t n_perms) { bpg::g_quad_threshold.store(n_perms); }
void bp_tune_poseidon_mx(int on) { bpg::g_poseidon_mx.store(on != 0); }
// Host only: the C-operand table of the matrix-core Poseidon kernels (poseidon_mx.cuh), 30 x 4 x 24 u32, for the CPU test
// that re-derives it from the round constants.
int bp_debug_poseidon_mx_cin(uint32_t* out) {
  if (!out) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_debug_poseidon_mx_cin: null buffer");
  static const poseidon::mx::CinTable t = poseidon::mx::make_cin_table();
  for (int i = 0; i < poseidon::mx::CIN_WORDS; i++) out[i] = t.v[i];
  return BP_OK;
}
/* 1 (default): the four-set matrix-core kernels take the partial rounds 4..19 in two groups of eight
 * (csrc/poseidon_mx.cuh, grp); 0: every round by itself.  Results are identical. */
/* The load-dependent choices (kernel forms, one-pass K5 / FRI combination): -1 (default) = by the number of
 * bp_generate_*_proof calls at work on the device (six or more = loaded); 0 / 1 = stated by a caller that drives the
 * L0 / L0.5 entry points from its own threads, or by a test that pins both paths. */
void bp_tune_merkle_wide(int log2_parents) { bpg::g_merkle_wide_log2.store(log2_parents < 8 || log2_parents > 24 ? 0 : log2_parents); }
void bp_tune_assume_loaded(int mode) { bpg::g_assume_loaded.store(mode < 0 ? -1 : (mode != 0)); }
// 0: every round by itself; 2: rounds 4..19 in two groups; 3 (or 1, the default): all 22 partial rounds in three
void bp_tune_poseidon_grouped(int mode) { bpg::g_poseidon_grouped.store(mode == 0 ? 0 : (mode == 2 ? 2 : 3)); }

// Host only: the operand images of one group as the device gets them (tests/test_mx_tables.py pins them to the
// integer model tools/poseidon_group_model.py).  out_ops: bp_debug_poseidon_group_ops(K) x 1024 bytes, out_cform: 64
// i32, out_cmain: 96 i32.
uint32_t bp_debug_poseidon_group_ops(uint32_t K) { return K >= 2 && K <= 8 ? (uint32_t)poseidon::group::layout((int)K).n_ops : 0; }
int bp_debug_poseidon_group_tables(uint32_t K, uint32_t r0, uint8_t* out_ops, int32_t* out_cform, int32_t* out_cmain,
                                   int32_t* out_max_plane_sum) try {
  if (!out_ops || !out_cform || !out_cmain) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_debug_poseidon_group_tables: null output");
  poseidon::group::Tables t;
  if (!poseidon::group::build((int)K, (int)r0, &t))
    return bpg::fail(BP_ERR_INVALID_INPUT, "bp_debug_poseidon_group_tables: no such group (K=%u r0=%u)", K, r0);
  std::memcpy(out_ops, t.ops.data(), t.ops.size());
  std::memcpy(out_cform, t.cform.data(), t.cform.size() * 4);
  std::memcpy(out_cmain, t.cmain.data(), t.cmain.size() * 4);
  if (out_max_plane_sum) *out_max_plane_sum = t.max_plane_sum;
  return BP_OK;
}
BPG_ABI_CATCH("bp_debug_poseidon_group_tables")

void bp_tune_poseidon_mx_sets(int sets) { bpg::g_mx_sets.store(sets == 1 || sets == 2 || sets == 4 ? sets : 0); }

uint64_t bp_merkle_digest_words(uint32_t log_leaves, uint32_t cap_height) {
  return (((uint64_t)2 << log_leaves) - ((uint64_t)1 << cap_height)) * 4;
}

int bp_poseidon_perm_batch(uint64_t* d_states, uint64_t n, void* stream) try {
  if (!n) return BP_OK;
  if (!d_states) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_poseidon_perm_batch: null buffer");
  hipStream_t st = bpg::as_stream(stream);
  using bpg::ceil_div;
  if (const int ns = bpg::mx_sets(n)) {
    BPG_MX_DISPATCH(ns, perm_batch_mx_kernel, n, 1, d_states, n)
  } else
    perm_batch_kernel<<<bpg::ceil_div(n, 256), 256, 0, st>>>(d_states, n);
  BPG_LAUNCH_CHECK();
  return BP_OK;
}
BPG_ABI_CATCH("bp_poseidon_perm_batch")

int bp_merkle_commit(const uint64_t* d_lde, uint64_t lde_stride, uint32_t n_cols, uint32_t log_n,
                     uint32_t rate_bits, uint32_t cap_height, uint64_t* d_digests, void* stream) try {
  return bpg::merkle_commit_cols(d_lde, lde_stride, n_cols, log_n, rate_bits, cap_height, d_digests,
                                 bpg::as_stream(stream), nullptr, nullptr, 1, 0, 0);
}
BPG_ABI_CATCH("bp_merkle_commit")

}  // extern "C"

namespace bpg {

int merkle_commit_cols(const uint64_t* d_lde, uint64_t lde_stride, uint32_t n_cols, uint32_t log_n, uint32_t rate_bits,
                       uint32_t cap_height, uint64_t* d_digests, hipStream_t st, uint64_t* mirror, bool* mirrored,
                       uint32_t batch, uint64_t lde_bstride, uint64_t dig_bstride) {
  const uint32_t log_leaves = log_n + rate_bits;
  if (!d_lde || !d_digests) return fail(BP_ERR_INVALID_INPUT, "bp_merkle_commit: null buffer");
  if (log_leaves > 30 || cap_height > log_leaves || n_cols == 0 || lde_stride < ((uint64_t)1 << log_leaves))
    return fail(BP_ERR_INVALID_INPUT,
                "bp_merkle_commit: bad shape (log_n=%u rate_bits=%u cap_height=%u n_cols=%u stride=%llu)", log_n,
                rate_bits, cap_height, n_cols, (unsigned long long)lde_stride);
  if (batch == 0 || batch > MAX_BATCH) return fail(BP_ERR_INVALID_INPUT, "merkle: batch of %u trees", batch);
  uint64_t rows = (uint64_t)1 << log_leaves;
  {
    // integer-ALU-bound family: the "bytes" slot of the profiler carries permutations
    KernelTimer kt(PROF_LEAF_HASH, st, n_cols > 4 ? (double)batch * (double)rows * (double)((n_cols + 7) / 8) : 0.0);
    // the matrix-core kernels take whole sets of 16 rows: a tree with fewer rows than a wave's sets takes fewer sets
    // (a forced set count -- bp_tune_poseidon_mx_sets, a low quad threshold -- must not read past a tiny matrix)
    int ns = mx_sets(rows * batch);
    while (ns > 1 && (uint64_t)16 * ns > rows) ns >>= 1;
    if (ns && rows >= 16) {
      BPG_MX_DISPATCH(ns, leaf_hash_mx_kernel, rows, batch, d_lde, lde_stride, n_cols, log_n, rate_bits, d_digests, lde_bstride, dig_bstride)
    } else if (rows * batch < quad_threshold())
      leaf_hash_quad_kernel<<<dim3(ceil_div(rows * 4, 256), 1, batch), 256, 0, st>>>(d_lde, lde_stride, n_cols, log_n, rate_bits,
                                                                               d_digests, lde_bstride, dig_bstride);
    else
      leaf_hash_kernel<<<dim3(ceil_div(rows, 256), 1, batch), 256, 0, st>>>(d_lde, lde_stride, n_cols, log_n, rate_bits, d_digests,
                                                                      lde_bstride, dig_bstride);
  }
  BPG_LAUNCH_CHECK();
  return merkle_upper_levels(d_digests, log_leaves, cap_height, st, mirror, mirrored, batch, dig_bstride);
}

}  // namespace bpg
