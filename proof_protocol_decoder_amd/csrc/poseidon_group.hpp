// poseidon_group.hpp -- operand tables of the GROUPED partial rounds of the matrix-core Poseidon kernels
// (poseidon_mx.cuh, "groups"; integer model and derivation: tools/poseidon_group_model.py).
//
// K partial rounds at a time (K <= 8): with t(r0) = (t0, w) the S-box-input form of the state at the group's first
// round and sigma_j = sbox(t0(r0 + j)), every later S-box input of the group is an affine form of w and the earlier
// sigmas, and so are the twelve words of t(r0 + K).  The coefficients are full 64-bit field elements; acting on the
// BYTES of w and sigma they become 8-row blocks of balanced base-256 digits, i.e. A operands of
// v_mfma_i32_16x16x64_i8.  Per round the kernel then recombines one word instead of twelve.
//
// Host side (this header + poseidon_group.cpp): builds the operand images once per process; hash_kernels.hip uploads
// them per device and the kernels copy them into LDS.
#pragma once
#include <cstdint>
#include <vector>

namespace poseidon {
namespace group {

constexpr int MAX_K = 8;
// operand index map for a group of K rounds (every operand: 64 lanes x 16 bytes = 1 KiB; lane l supplies
// A[row = l & 15][16 * (l >> 4) .. + 15]):
//   pair P (forms 4P .. 4P+3) is initialised at step j = 4P from
//     W(P, half, LO), W(P, half, HI) and, for P > 0, W(P, half, SIG) = the sigmas known by then (sigma_0 .. sigma_{4P-1})
//   and receives D(P, j, half) for every later sigma_j that a form of the pair still needs.
struct Layout {
  int K, n_pairs;
  int w_base[2];      // first W operand of pair P: order (L,LO) (L,HI) [(L,SIG)] (H,LO) (H,HI) [(H,SIG)]
  int w_per_half[2];  // 2 for P = 0, 3 for P = 1
  int d_base[2];      // first D operand of pair P: order (j, L), (j, H) for j = d_first[P] .. d_first[P] + d_count[P] - 1
  int d_first[2], d_count[2];
  int main_base;      // 18 operands: ((g * 2 + h) * 3 + chunk), chunk = LO, HI, SIG
  int n_ops;
};
constexpr Layout layout(int K) {
  Layout L{};
  L.K = K;
  L.n_pairs = (K + 3) / 4;
  int idx = 0;
  for (int P = 0; P < L.n_pairs; P++) {
    L.w_base[P] = idx;
    L.w_per_half[P] = P ? 3 : 2;
    idx += 2 * L.w_per_half[P];
    const int last_form = (4 * P + 3 < K - 1) ? 4 * P + 3 : K - 1;  // largest form index in the pair
    L.d_first[P] = 4 * P;
    L.d_count[P] = last_form - 4 * P;  // sigma_j for j = 4P .. last_form - 1
    if (L.d_count[P] < 0) L.d_count[P] = 0;
    L.d_base[P] = idx;
    idx += 2 * L.d_count[P];
  }
  L.main_base = idx;
  L.n_ops = idx + 18;
  return L;
}
constexpr int CFORM_WORDS = 2 * 2 * 16;  // [pair][half][row]   (row r -> lane group r >> 2, register r & 3)
constexpr int CMAIN_WORDS = 3 * 2 * 16;  // [g][h][row]

struct Tables {
  int K, r0;
  std::vector<uint8_t> ops;    // n_ops x 1024 bytes, device lane layout
  std::vector<int32_t> cform;  // CFORM_WORDS
  std::vector<int32_t> cmain;  // CMAIN_WORDS
  int32_t max_plane_sum;       // bound every plane sum stays under (checked < 2^23: mxa::planes)
};
// Builds the images for a group of K rounds whose first round is r0 (4 <= r0, r0 + K <= 26).  Throws nothing:
// returns false on an internal inconsistency (never expected).
bool build(int K, int r0, Tables* out);

}  // namespace group
}  // namespace poseidon
