// compact.cpp -- SURVEY.md section 8(f) row 1: the Erigon "compact block witness" decoder and the
// state-trie root it pins.  Host C++ on purpose: the reference's decoder is CPU parsing with a
// pointer-chasing structure and no data parallelism (protocol_decoder/src/compact/*,
// "not a resource bottleneck": README.md:9), so it stays beside the GPU path, not on it.
//
// Restates, from the reference sources:
//   byte stream -> instructions       compact_prestate_processing.rs:744-875 (opcodes :130-138,
//                                      account flags :885-893, raw 32-byte HASH :981-998)
//   key bytes -> nibbles              compact_prestate_processing.rs:1338-1390
//   collapse rules                    compact_prestate_processing.rs:387-462 (branch :464-527,
//                                      account leaf :537-606).  The reference sweeps a linked list
//                                      until no rule applies; every instruction consumes only
//                                      entries BEFORE it, so one left-to-right pass over a stack
//                                      applies the same rules in the same order.
//   node tree -> trie contents        compact_to_partial_trie.rs:49-165 (value leaves are
//                                      RLP-wrapped :119, account leaves are rlp(AccountRlp) :141-165,
//                                      constants protocol_decoder/src/types.rs:25-34)
// The trie itself is `eth_trie_utils` (git 7fc3c3f, not in /root/reference): its root hash is the
// Yellow Paper (appendix D) Merkle-Patricia hash, computed here directly on the witness node tree
// (equal to inserting every leaf / hash node for a well-formed witness).  Pinned by the six state
// roots and the instruction KAT the reference's own tests hold (tests/golden/).
#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>
#include "common.hpp"

namespace {

using Bytes = std::vector<uint8_t>;
using H256 = std::array<uint8_t, 32>;

// ---------------------------------------------------------------- Keccak-256 (FIPS 202 permutation, 0x01 padding)
void keccak_f(uint64_t st[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
      0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
      0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  for (int r = 0; r < 24; r++) {
    uint64_t bc[5];
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    uint64_t t = st[1];
    for (int i = 0; i < 24; i++) {
      int j = PIL[i];
      uint64_t b = st[j];
      st[j] = (t << ROT[i]) | (t >> (64 - ROT[i]));
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      uint64_t row[5];
      for (int i = 0; i < 5; i++) row[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~row[(i + 1) % 5]) & row[(i + 2) % 5];
    }
    st[0] ^= RC[r];
  }
}
H256 keccak256(const uint8_t* data, size_t len) {
  uint64_t st[25] = {0};
  const size_t rate = 136;
  uint8_t block[136];
  while (len >= rate) {
    for (size_t i = 0; i < rate / 8; i++) {
      uint64_t w;
      std::memcpy(&w, data + 8 * i, 8);
      st[i] ^= w;
    }
    keccak_f(st);
    data += rate;
    len -= rate;
  }
  std::memset(block, 0, rate);
  std::memcpy(block, data, len);
  block[len] ^= 0x01;
  block[rate - 1] ^= 0x80;
  for (size_t i = 0; i < rate / 8; i++) {
    uint64_t w;
    std::memcpy(&w, block + 8 * i, 8);
    st[i] ^= w;
  }
  keccak_f(st);
  H256 out;
  std::memcpy(out.data(), st, 32);
  return out;
}
H256 keccak256(const Bytes& b) { return keccak256(b.data(), b.size()); }

// ---------------------------------------------------------------- RLP
void rlp_len_prefix(Bytes& out, size_t len, uint8_t short_base) {
  if (len < 56) {
    out.push_back((uint8_t)(short_base + len));
  } else {
    uint8_t tmp[8];
    int n = 0;
    for (size_t l = len; l; l >>= 8) tmp[n++] = (uint8_t)l;
    out.push_back((uint8_t)(short_base + 55 + n));
    for (int i = n - 1; i >= 0; i--) out.push_back(tmp[i]);
  }
}
Bytes rlp_string(const uint8_t* d, size_t len) {
  Bytes out;
  if (len == 1 && d[0] < 0x80) {
    out.push_back(d[0]);
    return out;
  }
  rlp_len_prefix(out, len, 0x80);
  out.insert(out.end(), d, d + len);
  return out;
}
Bytes rlp_string(const Bytes& b) { return rlp_string(b.data(), b.size()); }
Bytes rlp_list(const std::vector<Bytes>& items) {
  size_t total = 0;
  for (auto& i : items) total += i.size();
  Bytes out;
  rlp_len_prefix(out, total, 0xc0);
  for (auto& i : items) out.insert(out.end(), i.begin(), i.end());
  return out;
}
// big-endian scalar with leading zeros stripped (U256 / u64 as RLP integers)
Bytes rlp_scalar_be(const Bytes& be) {
  size_t i = 0;
  while (i < be.size() && be[i] == 0) i++;
  return rlp_string(be.data() + i, be.size() - i);
}

// ---------------------------------------------------------------- instructions
using Nibbles = std::vector<uint8_t>;

enum class Op : uint8_t { Leaf = 0, Extension = 1, Branch = 2, Hash = 3, Code = 4, AccountLeaf = 5, EmptyRoot = 6 };

struct Instr {
  Op op;
  Nibbles key;
  Bytes key_bytes;   // raw key bytes as they appear in the stream (for the instruction listing)
  Bytes value;       // leaf value / code
  uint32_t mask = 0;
  H256 hash{};
  uint64_t nonce = 0;
  Bytes balance_be;  // big-endian, as read
  bool has_code = false, has_storage = false;
};

// compact_prestate_processing.rs:1338-1390
Nibbles key_bytes_to_nibbles(const Bytes& b) {
  Nibbles key;
  if (b.empty()) return key;
  if (b.size() == 1) key.push_back(b[0] & 0x0f);
  const bool is_odd = b[0] & 1;  // bit 1 (terminator) has no effect on the key
  if (b.size() == 1) return key;
  const size_t last = b.size() - 1;
  for (size_t i = 1; i < last; i++) {
    key.push_back(b[i] >> 4);
    key.push_back(b[i] & 0x0f);
  }
  key.push_back(b[last] >> 4);
  if (!is_odd) key.push_back(b[last] & 0x0f);
  return key;
}

struct Reader {
  const uint8_t* p;
  size_t len, pos = 0;
  std::string err;
  bool fail(const std::string& m) {
    if (err.empty()) err = m + " (byte position " + std::to_string(pos) + ")";
    return false;
  }
  bool byte(uint8_t* out) {
    if (pos >= len) return fail("Reached the end of the byte stream when we still expected more data");
    *out = p[pos++];
    return true;
  }
  bool cbor_head(uint8_t want_major, uint64_t* arg, const char* field) {
    uint8_t h;
    if (!byte(&h)) return false;
    if ((h >> 5) != want_major) return fail(std::string("Unable to parse the CBOR field \"") + field + "\": unexpected major type");
    const uint8_t ai = h & 0x1f;
    if (ai < 24) {
      *arg = ai;
      return true;
    }
    int n = ai == 24 ? 1 : ai == 25 ? 2 : ai == 26 ? 4 : ai == 27 ? 8 : 0;
    if (!n) return fail(std::string("Unable to parse the CBOR field \"") + field + "\": unsupported length encoding");
    uint64_t v = 0;
    for (int i = 0; i < n; i++) {
      uint8_t b;
      if (!byte(&b)) return false;
      v = (v << 8) | b;
    }
    *arg = v;
    return true;
  }
  bool cbor_bytes(Bytes* out, const char* field) {
    uint64_t n;
    if (!cbor_head(2, &n, field)) return false;
    if (n > len - pos) return fail(std::string("Unable to parse an expected byte vector (field name: ") + field + ")");
    out->assign(p + pos, p + pos + n);
    pos += n;
    return true;
  }
  bool cbor_uint(uint64_t* out, uint64_t max, const char* field) {
    if (!cbor_head(0, out, field)) return false;
    if (*out > max) return fail(std::string("Unable to parse the type for field \"") + field + "\": out of range");
    return true;
  }
};

// compact_prestate_processing.rs:683-697, 744-875
bool parse_instructions(const uint8_t* data, size_t len, uint8_t* version, std::vector<Instr>* out, std::string* err) {
  Reader r{data, len};
  if (!r.byte(version)) {
    *err = "Missing header";
    return false;
  }
  while (r.pos < r.len) {
    uint8_t opb;
    r.byte(&opb);
    if (opb > 6) {
      char buf[64];
      snprintf(buf, sizeof(buf), "Invalid opcode operator (\"%x\")", opb);
      *err = buf;
      return false;
    }
    Instr in;
    in.op = (Op)opb;
    bool ok = true;
    switch (in.op) {
      case Op::Leaf:
        ok = r.cbor_bytes(&in.key_bytes, "leaf key") && r.cbor_bytes(&in.value, "leaf value");
        break;
      case Op::Extension:
        ok = r.cbor_bytes(&in.key_bytes, "extension key");
        break;
      case Op::Branch: {
        uint64_t m = 0;
        ok = r.cbor_uint(&m, 0xFFFFFFFFull, "mask");
        in.mask = (uint32_t)m;
        break;
      }
      case Op::Hash:  // raw 32 bytes, not CBOR
        if (r.len - r.pos < 32) ok = r.fail("Unable to parse the type \"H256\" (field name: hash)");
        else {
          std::memcpy(in.hash.data(), r.p + r.pos, 32);
          r.pos += 32;
        }
        break;
      case Op::Code:
        ok = r.cbor_bytes(&in.value, "code");
        break;
      case Op::AccountLeaf: {
        uint8_t flags = 0;
        ok = r.cbor_bytes(&in.key_bytes, "account leaf key") && r.byte(&flags);
        if (!ok) break;
        in.has_code = flags & 1;
        in.has_storage = flags & 2;
        if (flags & 4) ok = r.cbor_uint(&in.nonce, ~0ull, "account leaf nonce");
        if (ok && (flags & 8)) ok = r.cbor_bytes(&in.balance_be, "account leaf balance");
        if (ok && in.balance_be.size() > 32) ok = r.fail("account leaf balance wider than 256 bits");
        uint64_t code_size;  // read and dropped, as the reference does
        if (ok && in.has_code) ok = r.cbor_uint(&code_size, ~0ull, "code size");
        break;
      }
      case Op::EmptyRoot:
        break;
    }
    if (!ok) {
      *err = r.err;
      return false;
    }
    in.key = key_bytes_to_nibbles(in.key_bytes);
    out->push_back(std::move(in));
  }
  return true;
}

// ---------------------------------------------------------------- node tree + Merkle-Patricia hashing
struct Node;
using NodeP = std::shared_ptr<Node>;
enum class Kind { Branch, Code, Empty, Hash, ValueLeaf, AccountLeaf, Extension };
struct Node {
  Kind kind;
  Nibbles key;            // leaf / extension
  Bytes payload;          // value leaf: raw value; code: bytes; account leaf: rlp(AccountRlp)
  H256 hash{};            // hash node
  NodeP child;            // extension
  NodeP children[16];     // branch
  uint32_t height = 1;    // nodes on the longest path below and including this one
};
// A state / storage path is at most 64 nibbles: at most 64 branches with an extension between any two,
// so no well-formed witness nests deeper than this.  The cap bounds every recursion over the tree
// (encode_node, collect_code, the shared_ptr destructor chain), whatever the client-supplied payload.
constexpr uint32_t MAX_TREE_HEIGHT = 160;

const H256 EMPTY_TRIE_HASH = {86, 232, 31, 23, 27, 204, 85, 166, 255, 131, 69, 230, 146, 192, 248, 110,
                              91, 72, 224, 27, 153, 108, 173, 192, 1, 98, 47, 181, 227, 99, 180, 33};
const H256 EMPTY_CODE_HASH = {197, 210, 70, 1, 134, 247, 35, 60, 146, 126, 125, 178, 220, 199, 3, 192,
                              229, 0, 182, 83, 202, 130, 39, 59, 123, 250, 216, 4, 93, 133, 164, 112};

Bytes hex_prefix(const Nibbles& k, bool leaf) {
  Bytes out;
  const uint8_t flag = (leaf ? 2 : 0) + (k.size() & 1);
  size_t i = 0;
  if (k.size() & 1) out.push_back((uint8_t)((flag << 4) | k[i++]));
  else out.push_back((uint8_t)(flag << 4));
  for (; i < k.size(); i += 2) out.push_back((uint8_t)((k[i] << 4) | k[i + 1]));
  return out;
}

struct Decoded {
  std::map<H256, Bytes> code;                 // code hash -> bytes
  std::set<H256> storage_roots;               // storage tries extracted (keyed by their root)
  std::vector<std::pair<Bytes, H256>> accounts;  // (key path nibbles, storage root) of every account leaf
};

// RLP of a trie node; `*is_hash_ref` set when the node is a hash node (its reference is the hash).
Bytes encode_node(const Node& n, Decoded* d, Nibbles& path);

// child reference inside a parent: raw RLP when shorter than 32 bytes, else keccak as a 32-byte string
Bytes node_ref(const NodeP& n, Decoded* d, Nibbles& path) {
  if (!n || n->kind == Kind::Empty) return Bytes{0x80};
  if (n->kind == Kind::Hash) return rlp_string(n->hash.data(), 32);
  Bytes enc = encode_node(*n, d, path);
  if (enc.size() < 32) return enc;
  H256 h = keccak256(enc);
  return rlp_string(h.data(), 32);
}
Bytes encode_node(const Node& n, Decoded* d, Nibbles& path) {
  switch (n.kind) {
    case Kind::Branch: {
      std::vector<Bytes> items;
      for (int i = 0; i < 16; i++) {
        path.push_back((uint8_t)i);
        items.push_back(node_ref(n.children[i], d, path));
        path.pop_back();
      }
      items.push_back(Bytes{0x80});  // branches carry no value in state / storage tries
      return rlp_list(items);
    }
    case Kind::Extension: {
      const size_t keep = path.size();
      path.insert(path.end(), n.key.begin(), n.key.end());
      Bytes child = node_ref(n.child, d, path);
      path.resize(keep);
      return rlp_list({rlp_string(hex_prefix(n.key, false)), child});
    }
    case Kind::ValueLeaf:  // compact_to_partial_trie.rs:119: the stored value is rlp(raw bytes)
      return rlp_list({rlp_string(hex_prefix(n.key, true)), rlp_string(rlp_string(n.payload))});
    case Kind::AccountLeaf: {
      if (d) {
        Bytes full(path.begin(), path.end());
        full.insert(full.end(), n.key.begin(), n.key.end());
        d->accounts.push_back({full, n.hash});
      }
      return rlp_list({rlp_string(hex_prefix(n.key, true)), rlp_string(n.payload)});
    }
    default:
      return Bytes{0x80};
  }
}
// HashedPartialTrie::hash(): keccak of the root encoding; a lone hash node is its own root
H256 trie_root(const NodeP& n, Decoded* d) {
  if (!n || n->kind == Kind::Empty || n->kind == Kind::Code) return EMPTY_TRIE_HASH;
  if (n->kind == Kind::Hash) return n->hash;
  Nibbles path;
  return keccak256(encode_node(*n, d, path));
}
void collect_code(const NodeP& n, Decoded* d) {  // Code nodes reachable in a (storage) subtree
  if (!n) return;
  if (n->kind == Kind::Code) d->code[keccak256(n->payload)] = n->payload;
  if (n->kind == Kind::Extension) collect_code(n->child, d);
  if (n->kind == Kind::Branch) for (auto& c : n->children) collect_code(c, d);
}

bool fail_msg(std::string* err, const std::string& m) {
  *err = m;
  return false;
}

// compact_prestate_processing.rs:387-606 as a stack machine
bool build_tree(const std::vector<Instr>& ins, Decoded* d, NodeP* root, std::string* err) {
  std::vector<NodeP> st;
  auto pop = [&](NodeP* out, const char* who) {
    if (st.empty()) return fail_msg(err, std::string("Invalid block witness entries: ") + who + " has no preceding node");
    *out = st.back();
    st.pop_back();
    return true;
  };
  for (const Instr& in : ins) {
    auto n = std::make_shared<Node>();
    switch (in.op) {
      case Op::EmptyRoot: n->kind = Kind::Empty; break;
      case Op::Hash: n->kind = Kind::Hash; n->hash = in.hash; break;
      case Op::Leaf: n->kind = Kind::ValueLeaf; n->key = in.key; n->payload = in.value; break;
      case Op::Code: n->kind = Kind::Code; n->payload = in.value; break;
      case Op::Extension:
        n->kind = Kind::Extension;
        n->key = in.key;
        if (!pop(&n->child, "Extension")) return false;
        if (n->child->kind == Kind::Extension)
          return fail_msg(err, "Invalid block witness entries: an extension node cannot have an extension child");
        n->height = n->child->height + 1;
        break;
      case Op::Branch: {
        n->kind = Kind::Branch;
        const int cnt = __builtin_popcount(in.mask);
        if ((int)st.size() < cnt) {
          char buf[160];
          snprintf(buf, sizeof(buf), "Branch mask %#x stated there should be %d preceding nodes but instead found %zu",
                   in.mask, cnt, st.size());
          return fail_msg(err, buf);
        }
        // the earliest of the `cnt` preceding nodes goes to the lowest set bit (:486-510)
        size_t idx = st.size() - cnt;
        for (int i = 0; i < 16; i++)
          if (in.mask & (1u << i)) {
            n->children[i] = st[idx++];
            n->height = std::max(n->height, n->children[i]->height + 1);
          }
        if (in.mask >> 16) return fail_msg(err, "Branch mask has bits above the 16 children");
        st.resize(st.size() - cnt);
        break;
      }
      case Op::AccountLeaf: {
        n->kind = Kind::AccountLeaf;
        n->key = in.key;
        H256 storage_root = EMPTY_TRIE_HASH, code_hash = EMPTY_CODE_HASH;
        if (in.has_storage) {  // nearest preceding node is the storage-trie root (:537-606)
          NodeP s;
          if (!pop(&s, "AccountLeaf (storage)")) return false;
          if (s->kind == Kind::Code) return fail_msg(err, "Invalid block witness entries: a code node cannot be a storage root");
          storage_root = trie_root(s, nullptr);
          d->storage_roots.insert(storage_root);
          collect_code(s, d);
        }
        if (in.has_code) {     // then the code (bytes or hash)
          NodeP c;
          if (!pop(&c, "AccountLeaf (code)")) return false;
          if (c->kind == Kind::Code) {
            code_hash = keccak256(c->payload);
            d->code[code_hash] = c->payload;
          } else if (c->kind == Kind::Hash) {
            code_hash = c->hash;
          } else {
            return fail_msg(err, "Invalid block witness entries: account code must be a code or a hash node");
          }
        }
        Bytes nonce_be(8);
        for (int i = 0; i < 8; i++) nonce_be[i] = (uint8_t)(in.nonce >> (56 - 8 * i));
        n->payload = rlp_list({rlp_scalar_be(nonce_be), rlp_scalar_be(in.balance_be), rlp_string(storage_root.data(), 32),
                               rlp_string(code_hash.data(), 32)});
        n->hash = storage_root;  // remembered for the account list
        break;
      }
    }
    if (n->height > MAX_TREE_HEIGHT)
      return fail_msg(err, "Invalid block witness entries: node tree nests deeper than any 64-nibble path allows");
    st.push_back(n);
  }
  if (st.size() > 1) {
    return fail_msg(err, "There were multiple entries remaining after the compact block witness was processed (" +
                             std::to_string(st.size()) + " remaining)");
  }
  *root = st.empty() ? nullptr : st[0];
  return true;
}

std::string hex(const uint8_t* d, size_t n) {
  static const char* H = "0123456789abcdef";
  std::string s;
  for (size_t i = 0; i < n; i++) {
    s.push_back(H[d[i] >> 4]);
    s.push_back(H[d[i] & 15]);
  }
  return s;
}

}  // namespace

extern "C" {

void bp_keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
  H256 h = keccak256(data, len);
  std::memcpy(out, h.data(), 32);
}

// process_compact_prestate (compact_prestate_processing.rs:1240-1281): header version, state root,
// and the sizes of what was extracted.  Any pointer but `witness` may be NULL.
int bp_compact_decode(const uint8_t* witness, size_t len, uint8_t* header_version, uint8_t state_root[32],
                      uint32_t* n_accounts, uint32_t* n_storage_tries, uint32_t* n_code,
                      uint32_t* n_accounts_missing_storage) try {
  if (!witness && len) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_compact_decode: null witness");
  uint8_t ver = 0;
  std::vector<Instr> ins;
  std::string err;
  if (!parse_instructions(witness, len, &ver, &ins, &err)) return bpg::fail(BP_ERR_INVALID_INPUT, "%s", err.c_str());
  Decoded d;
  NodeP root;
  if (!build_tree(ins, &d, &root, &err)) return bpg::fail(BP_ERR_INVALID_INPUT, "%s", err.c_str());
  if (root && root->kind == Kind::Code) d.code[keccak256(root->payload)] = root->payload;
  H256 sr = trie_root(root, &d);
  if (header_version) *header_version = ver;
  if (state_root) std::memcpy(state_root, sr.data(), 32);
  if (n_accounts) *n_accounts = (uint32_t)d.accounts.size();
  if (n_storage_tries) *n_storage_tries = (uint32_t)d.storage_roots.size();
  if (n_code) *n_code = (uint32_t)d.code.size();
  if (n_accounts_missing_storage) {  // complex_test_payloads.rs:73-90: every non-empty root must have its trie
    uint32_t miss = 0;
    for (auto& a : d.accounts)
      if (a.second != EMPTY_TRIE_HASH && !d.storage_roots.count(a.second)) miss++;
    *n_accounts_missing_storage = miss;
  }
  return BP_OK;
}
BPG_ABI_CATCH("bp_compact_decode")

// Instruction listing, one per line ("leaf <key nibbles hex> <value hex>", "branch <mask>", ...), for the
// instruction-level KAT (compact_prestate_processing.rs:1471-1497).  Release with bp_free_buffer.
int bp_compact_instructions(const uint8_t* witness, size_t len, uint8_t** text_out, size_t* text_len) try {
  if ((!witness && len) || !text_out || !text_len) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_compact_instructions: null argument");
  uint8_t ver = 0;
  std::vector<Instr> ins;
  std::string err;
  if (!parse_instructions(witness, len, &ver, &ins, &err)) return bpg::fail(BP_ERR_INVALID_INPUT, "%s", err.c_str());
  std::string s;
  auto nib = [](const Nibbles& k) {
    std::string t;
    for (uint8_t n : k) t.push_back("0123456789abcdef"[n]);
    return t.empty() ? std::string("-") : t;
  };
  for (auto& in : ins) {
    switch (in.op) {
      case Op::Leaf: s += "leaf " + nib(in.key) + " " + hex(in.value.data(), in.value.size()); break;
      case Op::Extension: s += "extension " + nib(in.key); break;
      case Op::Branch: s += "branch " + std::to_string(in.mask); break;
      case Op::Hash: s += "hash " + hex(in.hash.data(), 32); break;
      case Op::Code: s += "code " + hex(in.value.data(), in.value.size()); break;
      case Op::AccountLeaf:
        s += "account_leaf " + nib(in.key) + " nonce=" + std::to_string(in.nonce) + " balance=" +
             (in.balance_be.empty() ? std::string("0") : hex(in.balance_be.data(), in.balance_be.size())) +
             " code=" + (in.has_code ? "1" : "0") + " storage=" + (in.has_storage ? "1" : "0");
        break;
      case Op::EmptyRoot: s += "empty_root"; break;
    }
    s.push_back('\n');
  }
  *text_len = s.size();
  *text_out = static_cast<uint8_t*>(std::malloc(s.size() + 1));
  if (!*text_out) return bpg::fail(BP_ERR_DEVICE, "host allocation failed");
  std::memcpy(*text_out, s.data(), s.size());
  (*text_out)[s.size()] = 0;
  return BP_OK;
}
BPG_ABI_CATCH("bp_compact_instructions")

}  // extern "C"
