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
#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "common.hpp"
#include "compact.hpp"

namespace {

using Bytes = mpt::Bytes;
using H256 = mpt::H256;
using mpt::EMPTY_CODE_HASH;
using mpt::EMPTY_TRIE_HASH;
using mpt::keccak256;
using mpt::rlp_list;
using mpt::rlp_scalar_be;
using mpt::rlp_string;

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

// ---------------------------------------------------------------- witness node tree
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
// (to_trie_node, collect_code, the trie code of mpt.cpp, the shared_ptr destructor chain), whatever the
// client-supplied payload.
constexpr uint32_t MAX_TREE_HEIGHT = 160;

// witness node tree -> trie nodes (compact_to_partial_trie.rs:49-165): value leaves store rlp(value) (:119),
// account leaves rlp(AccountRlp) (:141-165); code nodes and the empty root are not trie nodes.  The reference
// re-inserts every leaf / hash node into an empty trie; for a well-formed witness the tree IS that trie.
mpt::NodeP to_trie_node(const NodeP& n) {
  if (!n) return mpt::make_empty();
  switch (n->kind) {
    case Kind::Branch: {
      mpt::NodeP ch[16];
      for (int i = 0; i < 16; i++) ch[i] = to_trie_node(n->children[i]);
      return mpt::make_branch(ch);
    }
    case Kind::Extension: return mpt::make_extension(n->key, to_trie_node(n->child));
    case Kind::Hash: return mpt::make_hash(n->hash);
    case Kind::ValueLeaf: return mpt::make_leaf(n->key, rlp_string(n->payload));
    case Kind::AccountLeaf: return mpt::make_leaf(n->key, n->payload);
    default: return mpt::make_empty();
  }
}
void collect_code(const NodeP& n, std::map<H256, Bytes>* code) {  // Code nodes reachable in a (storage) subtree
  if (!n) return;
  if (n->kind == Kind::Code) (*code)[keccak256(n->payload)] = n->payload;
  if (n->kind == Kind::Extension) collect_code(n->child, code);
  if (n->kind == Kind::Branch) for (auto& c : n->children) collect_code(c, code);
}

bool fail_msg(std::string* err, const std::string& m) {
  *err = m;
  return false;
}

// compact_prestate_processing.rs:387-606 as a stack machine
bool build_tree(const std::vector<Instr>& ins, bpg::CompactOut* d, NodeP* root, std::string* err) {
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
          mpt::Trie storage(to_trie_node(s));
          storage_root = storage.hash();
          d->storage_by_root[storage_root] = storage;  // keyed by root first (compact_prestate_processing.rs:617-619)
          collect_code(s, &d->code);
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

namespace bpg {

// h_addr_nibs_to_h256 (protocol_decoder/src/utils.rs:47-58): left-pad a short path with zero bytes
static H256 nibbles_to_h256(const mpt::Nibbles& k) {
  Bytes b;
  size_t i = 0;
  if (k.size() & 1) b.push_back(k[i++]);
  for (; i + 1 < k.size() + 1 && i < k.size(); i += 2) b.push_back((uint8_t)((k[i] << 4) | k[i + 1]));
  H256 h{};
  if (b.size() > 32) b.erase(b.begin(), b.end() - 32);
  std::memcpy(h.data() + 32 - b.size(), b.data(), b.size());
  return h;
}

// process_compact_prestate (compact_prestate_processing.rs:1240-1281) + the re-keying of
// compact_to_partial_trie.rs:167-190 (storage tries keyed by hashed account address).
bool decode_compact(const uint8_t* witness, size_t len, CompactOut* out, std::string* err) {
  std::vector<Instr> ins;
  if (!parse_instructions(witness, len, &out->header_version, &ins, err)) return false;
  NodeP root;
  if (!build_tree(ins, out, &root, err)) return false;
  if (root && root->kind == Kind::Code) out->code[keccak256(root->payload)] = root->payload;
  out->state = mpt::Trie(to_trie_node(root));
  std::vector<mpt::Item> items;
  out->state.items(&items);
  for (auto& it : items) {
    if (it.is_hash) continue;
    mpt::Account acc;
    if (!mpt::account_decode(it.value, &acc)) continue;  // a value leaf in the state trie: not an account
    const H256 h_addr = nibbles_to_h256(it.path);
    out->accounts.push_back({h_addr, acc});
    auto st = out->storage_by_root.find(acc.storage_root);
    if (st != out->storage_by_root.end()) out->storage[h_addr] = st->second;
  }
  return true;
}

}  // namespace bpg

extern "C" {

void bp_keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
  H256 h = keccak256(data, len);
  std::memcpy(out, h.data(), 32);
}

// Keccak-256 of `data` together with the state that goes into each of its permutations (len / 136 + 1 of them, 25
// lanes each, lane index x + 5y): what the Keccak table of a transaction that hashes `data` has to contain
// (bp_generate_txn_proof_keccak, bp_keccak_trace).  states_out may be NULL to count.
int bp_keccak256_permutation_inputs(const uint8_t* data, size_t len, uint8_t digest_out[32], uint64_t* states_out,
                                    size_t max_perms, size_t* n_perms_out) try {
  if ((!data && len) || !n_perms_out) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_keccak256_permutation_inputs: null argument");
  std::vector<uint64_t> st;
  const H256 h = mpt::keccak256_traced(data, len, &st);
  *n_perms_out = st.size() / 25;
  if (digest_out) std::memcpy(digest_out, h.data(), 32);
  if (states_out) {
    if (st.size() / 25 > max_perms) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_keccak256_permutation_inputs: %zu permutations, room for %zu", st.size() / 25, max_perms);
    std::memcpy(states_out, st.data(), st.size() * 8);
  }
  return BP_OK;
}
BPG_ABI_CATCH("bp_keccak256_permutation_inputs")

// The same hash as rows of the Keccak sponge table (AIR 6; bp_keccak_sponge_trace, bp_generate_txn_proof_witness): one
// row of 44 words per 136-byte block -- flags (1 full, 2 final), message bytes in the block, the block as absorbed (17
// words), the 25 lanes of the state before it.  rows_out may be NULL to count.
int bp_keccak256_sponge_rows(const uint8_t* data, size_t len, uint8_t digest_out[32], uint64_t* rows_out, size_t max_rows,
                             size_t* n_rows_out) try {
  if ((!data && len) || !n_rows_out) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_keccak256_sponge_rows: null argument");
  std::vector<uint64_t> rows;
  const H256 h = mpt::keccak256_sponge_rows(data, len, &rows);
  *n_rows_out = rows.size() / 44;
  if (digest_out) std::memcpy(digest_out, h.data(), 32);
  if (rows_out) {
    if (rows.size() / 44 > max_rows) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_keccak256_sponge_rows: %zu rows, room for %zu", rows.size() / 44, max_rows);
    std::memcpy(rows_out, rows.data(), rows.size() * 8);
  }
  return BP_OK;
}
BPG_ABI_CATCH("bp_keccak256_sponge_rows")

// process_compact_prestate (compact_prestate_processing.rs:1240-1281): header version, state root,
// and the sizes of what was extracted.  Any pointer but `witness` may be NULL.
int bp_compact_decode(const uint8_t* witness, size_t len, uint8_t* header_version, uint8_t state_root[32],
                      uint32_t* n_accounts, uint32_t* n_storage_tries, uint32_t* n_code,
                      uint32_t* n_accounts_missing_storage) try {
  if (!witness && len) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_compact_decode: null witness");
  bpg::CompactOut d;
  std::string err;
  if (!bpg::decode_compact(witness, len, &d, &err)) return bpg::fail(BP_ERR_INVALID_INPUT, "%s", err.c_str());
  const H256 sr = d.state.hash();
  if (header_version) *header_version = d.header_version;
  if (state_root) std::memcpy(state_root, sr.data(), 32);
  if (n_accounts) *n_accounts = (uint32_t)d.accounts.size();
  if (n_storage_tries) *n_storage_tries = (uint32_t)d.storage_by_root.size();
  if (n_code) *n_code = (uint32_t)d.code.size();
  if (n_accounts_missing_storage) {  // complex_test_payloads.rs:73-90: every non-empty root must have its trie
    uint32_t miss = 0;
    for (auto& a : d.accounts)
      if (a.second.storage_root != EMPTY_TRIE_HASH && !d.storage_by_root.count(a.second.storage_root)) miss++;
    *n_accounts_missing_storage = miss;
  }
  return BP_OK;
}
BPG_ABI_CATCH("bp_compact_decode")

// The reference's full output (ProcessedCompactOutput{header, witness_out{tries{state, storage}, code}},
// compact_prestate_processing.rs:1243-1281) in the byte layout of DESIGN.md section 8c:
//   "BPGCWIT1" | version:u8 | state: len:u32 trie | n_storage:u32 (h_addr[32] len:u32 trie)* | n_code:u32 (hash[32] len:u32 bytes)*
// storage tries keyed by HASHED ACCOUNT ADDRESS (after the re-keying of compact_to_partial_trie.rs:167-190), both
// lists in ascending key order; tries in the node format of mpt.cpp.  Release with bp_free_buffer.
int bp_compact_decode_full(const uint8_t* witness, size_t len, uint8_t** out, size_t* out_len) try {
  if ((!witness && len) || !out || !out_len) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_compact_decode_full: null argument");
  bpg::CompactOut d;
  std::string err;
  if (!bpg::decode_compact(witness, len, &d, &err)) return bpg::fail(BP_ERR_INVALID_INPUT, "%s", err.c_str());
  Bytes o;
  const char magic[8] = {'B', 'P', 'G', 'C', 'W', 'I', 'T', '1'};
  o.insert(o.end(), magic, magic + 8);
  o.push_back(d.header_version);
  bpg::put_blob(&o, bpg::trie_bytes(d.state));
  bpg::put_u32(&o, (uint32_t)d.storage.size());
  for (auto& kv : d.storage) {
    o.insert(o.end(), kv.first.begin(), kv.first.end());
    bpg::put_blob(&o, bpg::trie_bytes(kv.second));
  }
  bpg::put_u32(&o, (uint32_t)d.code.size());
  for (auto& kv : d.code) {
    o.insert(o.end(), kv.first.begin(), kv.first.end());
    bpg::put_blob(&o, kv.second);
  }
  return bpg::emit_bytes(o, out, out_len);
}
BPG_ABI_CATCH("bp_compact_decode_full")

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
