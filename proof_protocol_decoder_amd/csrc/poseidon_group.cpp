// poseidon_group.cpp -- see poseidon_group.hpp.  Host code only (exact integer arithmetic mod p with __int128);
// restates tools/poseidon_group_model.py, which tests/test_mx_tables.py compares it with byte for byte.
#include "poseidon_group.hpp"
#include <array>
#include <map>
#include <tuple>

namespace poseidon {
namespace group {
namespace {

constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
constexpr uint64_t RC[360] = {
#include "poseidon_rc.inc"
};
inline uint64_t addm(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + b) % P); }
inline uint64_t mulm(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % P); }
inline uint32_t mds(int i, int k) {
  static const uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  return C[(k - i + 12) % 12] + ((i | k) == 0 ? 8u : 0u);
}

constexpr int NV = 11 + MAX_K;  // variables: w_1 .. w_11, sigma_0 .. sigma_{K-1}
struct Affine {
  std::array<uint64_t, NV> coef{};
  uint64_t constant = 0;
  bool used = false;
};

// forms[j] (j = 1 .. K-1; forms[0] unused: F_0 = t0 itself) and the twelve words of t(r0 + K)
void affine_group(int K, int r0, std::vector<Affine>& forms, std::vector<Affine>& last) {
  std::vector<Affine> w(11);
  for (int i = 0; i < 11; i++) { w[i].coef[i] = 1; w[i].used = true; }
  forms.assign(K, Affine{});
  for (int j = 0; j < K; j++) {
    std::vector<Affine> t(12);
    t[0].coef[11 + j] = 1;
    t[0].used = true;
    for (int i = 1; i < 12; i++) t[i] = w[i - 1];
    std::vector<Affine> nxt(12);
    for (int i = 0; i < 12; i++) {
      nxt[i].used = true;
      for (int k = 0; k < 12; k++) {
        const uint64_t m = mds(i, k);
        for (int v = 0; v < NV; v++) nxt[i].coef[v] = addm(nxt[i].coef[v], mulm(m, t[k].coef[v]));
        nxt[i].constant = addm(nxt[i].constant, mulm(m, t[k].constant));
      }
      if (r0 + j + 1 < 30) nxt[i].constant = addm(nxt[i].constant, RC[(r0 + j + 1) * 12 + i]);
    }
    if (j + 1 < K) forms[j + 1] = nxt[0];
    for (int i = 0; i < 11; i++) w[i] = nxt[i + 1];
    last = nxt;
  }
}

// d in [0, p) -> eight digits in [-128, 127] of d or d - p, least significant first
bool balanced_digits(uint64_t d, int8_t out[8]) {
  __int128 v = d <= P / 2 ? (__int128)d : (__int128)d - (__int128)P;
  for (int i = 0; i < 8; i++) {
    int r = (int)(((v + 128) % 256 + 256) % 256) - 128;
    out[i] = (int8_t)r;
    v = (v - r) / 256;
  }
  return v == 0;
}

enum Chunk { LO = 0, HI = 1, SIG = 2 };
// operand byte k of a chunk -> variable index and byte q, or false
bool var_of_k(Chunk c, int k, int K, int* v, int* q) {
  const int kb = k >> 4, dword = (k >> 2) & 3, byte = k & 3;
  if (c != SIG) {
    if (dword == 3) return false;
    const int word = kb + 4 * dword;
    if (word == 0) return false;  // word 0 holds t0, which only feeds the S-box
    *v = word - 1;
    *q = byte + (c == HI ? 4 : 0);
    return true;
  }
  const int j = kb + 4 * (dword >> 1);
  if (j >= K) return false;
  *v = 11 + j;
  *q = byte + 4 * (dword & 1);
  return true;
}

struct Tile {
  int target[16], plane[16];  // what each row computes (target < 0: unused row)
  int8_t A[3][16][64];
  int32_t C[16];
};

struct Builder {
  int K;
  const std::vector<Affine>* exprs;
  std::map<std::tuple<int, int, int>, std::array<int8_t, 8>> digs;
  bool ok = true;
  void fill(Tile& t) {
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 16; r++)
        for (int k = 0; k < 64; k++) {
          t.A[c][r][k] = 0;
          const int tg = t.target[r];
          if (tg < 0 || tg >= (int)exprs->size() || !(*exprs)[tg].used) continue;
          int v, q;
          if (!var_of_k((Chunk)c, k, K, &v, &q)) continue;
          const uint64_t co = (*exprs)[tg].coef[v];
          if (!co) continue;
          auto key = std::make_tuple(tg, v, q);
          auto it = digs.find(key);
          if (it == digs.end()) {
            std::array<int8_t, 8> d{};
            uint64_t sh = co;
            for (int s = 0; s < 8 * q; s++) sh = addm(sh, sh);  // co * 2^(8q) mod p
            ok = balanced_digits(sh, d.data()) && ok;
            it = digs.emplace(key, d).first;
          }
          t.A[c][r][k] = it->second[t.plane[r]];
        }
  }
};

// C_p = 128 * rowsum + (-lo_p) + cdigit_p: undoes the -128 of the byte operands, lifts every plane sum to >= 0 and
// carries the form's constant; sum_p (C_p + sum A (x - 128)) 256^p == form (mod p)
int32_t finish_constants(const std::vector<Affine>& exprs, std::vector<Tile>& tiles) {
  std::map<std::pair<int, int>, int64_t> lo, rowsum, pos;
  for (auto& t : tiles)
    for (int r = 0; r < 16; r++) {
      if (t.target[r] < 0) continue;
      int64_t neg = 0, sum = 0, ps = 0;
      for (int c = 0; c < 3; c++)
        for (int k = 0; k < 64; k++) {
          const int a = t.A[c][r][k];
          sum += a;
          if (a < 0) neg += a; else ps += a;
        }
      lo[{t.target[r], t.plane[r]}] = 255 * neg;
      rowsum[{t.target[r], t.plane[r]}] = sum;
      pos[{t.target[r], t.plane[r]}] = ps;
    }
  int64_t bound = 0;
  for (auto& t : tiles)
    for (int r = 0; r < 16; r++) {
      t.C[r] = 0;
      const int tg = t.target[r];
      if (tg < 0 || tg >= (int)exprs.size() || !exprs[tg].used) continue;
      uint64_t shift = 0;
      for (int p = 7; p >= 0; p--) {
        for (int s = 0; s < 8; s++) shift = addm(shift, shift);
        shift = addm(shift, (uint64_t)(-lo[{tg, p}]) % P);
      }
      const uint64_t cd = addm(exprs[tg].constant, P - shift);
      const int64_t cdig = (cd >> (8 * t.plane[r])) & 0xFF;
      const auto key = std::make_pair(tg, t.plane[r]);
      t.C[r] = (int32_t)(128 * rowsum[key] - lo[key] + cdig);
      const int64_t hi = 255 * pos[key] - lo[key] + 255;
      if (hi > bound) bound = hi;
    }
  return (int32_t)bound;
}

void emit_operand(const int8_t (&A)[16][64], int j_only, int j_below, int K, std::vector<uint8_t>& ops) {
  // lane l = (row = l & 15, kblock = l >> 4) holds bytes A[row][16 kblock .. 16 kblock + 15].
  // j_only >= 0: keep only sigma_{j_only}'s bytes (a delta operand); j_below >= 0: only sigma_j, j < j_below.
  for (int l = 0; l < 64; l++)
    for (int b = 0; b < 16; b++) {
      const int k = 16 * (l >> 4) + b;
      int8_t a = A[l & 15][k];
      if (j_only >= 0 || j_below >= 0) {
        int v, q;
        const bool is = var_of_k(SIG, k, K, &v, &q);
        const int j = is ? v - 11 : -1;
        if (!is || (j_only >= 0 && j != j_only) || (j_below >= 0 && j >= j_below)) a = 0;
      }
      ops.push_back((uint8_t)a);
    }
}

}  // namespace

bool build(int K, int r0, Tables* out) {
  if (K < 2 || K > MAX_K || r0 < 4 || r0 + K > 26) return false;
  std::vector<Affine> forms, last;
  affine_group(K, r0, forms, last);
  const Layout L = layout(K);
  std::vector<Tile> ft(2 * L.n_pairs), mt(6);
  Builder fb{K, &forms}, mb{K, &last};
  for (int Pi = 0; Pi < L.n_pairs; Pi++)
    for (int half = 0; half < 2; half++) {
      Tile& t = ft[2 * Pi + half];
      for (int r = 0; r < 16; r++) {
        const int f = 4 * Pi + (r >> 2);
        t.target[r] = (f >= 1 && f < K) ? f : -1;
        t.plane[r] = 4 * half + (r & 3);
      }
      fb.fill(t);
    }
  for (int g = 0; g < 3; g++)
    for (int h = 0; h < 2; h++) {
      Tile& t = mt[2 * g + h];
      for (int r = 0; r < 16; r++) {
        t.target[r] = (r >> 2) + 4 * g;
        t.plane[r] = 4 * h + (r & 3);
      }
      mb.fill(t);
    }
  if (!fb.ok || !mb.ok) return false;
  const int32_t b1 = finish_constants(forms, ft), b2 = finish_constants(last, mt);
  out->K = K;
  out->r0 = r0;
  out->max_plane_sum = b1 > b2 ? b1 : b2;
  out->ops.clear();
  out->ops.reserve((size_t)L.n_ops * 1024);
  for (int Pi = 0; Pi < L.n_pairs; Pi++) {
    for (int half = 0; half < 2; half++) {
      const Tile& t = ft[2 * Pi + half];
      emit_operand(t.A[LO], -1, -1, K, out->ops);
      emit_operand(t.A[HI], -1, -1, K, out->ops);
      if (Pi) emit_operand(t.A[SIG], -1, 4 * Pi, K, out->ops);
    }
    for (int d = 0; d < L.d_count[Pi]; d++)
      for (int half = 0; half < 2; half++) emit_operand(ft[2 * Pi + half].A[SIG], L.d_first[Pi] + d, -1, K, out->ops);
  }
  for (int g = 0; g < 3; g++)
    for (int h = 0; h < 2; h++)
      for (int c = 0; c < 3; c++) emit_operand(mt[2 * g + h].A[c], -1, -1, K, out->ops);
  if ((int)out->ops.size() != L.n_ops * 1024) return false;
  out->cform.assign(CFORM_WORDS, 0);
  for (int Pi = 0; Pi < L.n_pairs; Pi++)
    for (int half = 0; half < 2; half++)
      for (int r = 0; r < 16; r++) out->cform[(Pi * 2 + half) * 16 + r] = ft[2 * Pi + half].C[r];
  out->cmain.assign(CMAIN_WORDS, 0);
  for (int t = 0; t < 6; t++)
    for (int r = 0; r < 16; r++) out->cmain[t * 16 + r] = mt[t].C[r];
  return out->max_plane_sum < (1 << 23);
}

}  // namespace group
}  // namespace poseidon
