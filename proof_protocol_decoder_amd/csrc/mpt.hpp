// mpt.hpp -- host-side Merkle-Patricia partial trie for the decoder rows of SURVEY.md section 8(f): what
// `eth_trie_utils::partial_trie::HashedPartialTrie` (git 7fc3c3f, NOT in /root/reference) gives the reference's
// protocol_decoder: insert / get / delete, root hash, items(), and `create_trie_subset` (keep the paths of some keys,
// hash out everything else).  Restated from the Yellow Paper (appendix D) and from how the reference uses the
// library (protocol_decoder/src/decoding.rs:179-292, 404-428; compact/compact_to_partial_trie.rs:49-190).
// CPU only by design: trie bookkeeping is pointer chasing with no data parallelism ("not a resource bottleneck",
// README.md:9).  Nodes are immutable and shared, so cloning a trie (the reference clones them freely,
// decoding.rs:85-96) is one pointer copy and every update rebuilds only the path it touches.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace mpt {

using Bytes = std::vector<uint8_t>;
using H256 = std::array<uint8_t, 32>;
using Nibbles = std::vector<uint8_t>;  // one nibble per element

H256 keccak256(const uint8_t* data, size_t len);
H256 keccak256_traced(const uint8_t* data, size_t len, std::vector<uint64_t>* perm_inputs);  // + the permutation inputs
H256 keccak256_sponge_rows(const uint8_t* data, size_t len, std::vector<uint64_t>* rows);     // + the sponge table's rows
inline H256 keccak256(const Bytes& b) { return keccak256(b.data(), b.size()); }

// protocol_decoder/src/types.rs:25-43
extern const H256 EMPTY_TRIE_HASH;  // keccak(rlp(""))
extern const H256 EMPTY_CODE_HASH;  // keccak("")

// ---------------------------------------------------------------- RLP
Bytes rlp_string(const uint8_t* d, size_t len);
inline Bytes rlp_string(const Bytes& b) { return rlp_string(b.data(), b.size()); }
Bytes rlp_list(const std::vector<Bytes>& items);
Bytes rlp_scalar_be(const Bytes& be);  // big-endian integer, leading zeros stripped
Bytes rlp_u64(uint64_t v);
struct RlpItem {
  bool is_list = false;
  const uint8_t* payload = nullptr;
  size_t len = 0;    // payload length
  size_t total = 0;  // header + payload
};
bool rlp_parse(const uint8_t* d, size_t n, RlpItem* out);              // the one item starting at d
bool rlp_children(const RlpItem& list, std::vector<RlpItem>* out);     // items of a list payload

// AccountRlp (plonky2_evm::generation::mpt, used at decoding.rs:236-262, compact_to_partial_trie.rs:141-165):
// rlp([nonce, balance, storage_root, code_hash]); nonce and balance are U256 kept as stripped big-endian bytes.
struct Account {
  Bytes nonce_be, balance_be;
  H256 storage_root, code_hash;
};
bool account_decode(const Bytes& rlp, Account* out);
Bytes account_encode(const Account& a);
Bytes u256_add_be(const Bytes& a, const Bytes& b, bool* overflow);  // stripped big-endian sum

inline Nibbles nibbles_of_bytes(const uint8_t* d, size_t n) {  // Nibbles::from_bytes_be / from_h256_be
  Nibbles k(2 * n);
  for (size_t i = 0; i < n; i++) {
    k[2 * i] = d[i] >> 4;
    k[2 * i + 1] = d[i] & 15;
  }
  return k;
}
inline Nibbles nibbles_of(const H256& h) { return nibbles_of_bytes(h.data(), 32); }

// ---------------------------------------------------------------- trie
enum class Kind : uint8_t { Empty = 0, Hash = 1, Branch = 2, Extension = 3, Leaf = 4 };
struct Node;
using NodeP = std::shared_ptr<const Node>;
struct Node {
  Kind kind = Kind::Empty;
  Nibbles key;         // extension / leaf
  Bytes value;         // leaf; branch value (unused by state / storage / txn / receipt tries: RLP keys are prefix-free)
  H256 hash{};         // hash node
  NodeP child;         // extension
  NodeP children[16];  // branch (null = empty)
  mutable Bytes enc;   // cached RLP encoding (nodes are immutable)
  mutable bool enc_valid = false;
};

enum class Status { Ok = 0, HitHashNode, CannotCollapse, Malformed };
const char* status_text(Status s);

struct Item {
  Nibbles path;
  bool is_hash;
  Bytes value;  // leaf value, or the 32 hash bytes
};

class Trie {
 public:
  Trie();
  explicit Trie(NodeP root) : root_(std::move(root)) {}
  static Trie of_hash(const H256& h);  // HashedPartialTrie::new(Node::Hash(h)), decoding.rs:603
  const NodeP& root() const { return root_; }
  bool is_empty() const { return root_->kind == Kind::Empty; }

  Status insert(const Nibbles& k, const Bytes& v);
  // *out = nullptr when the key is absent
  Status get(const Nibbles& k, const Bytes** out) const;
  Status remove(const Nibbles& k, bool* existed);
  H256 hash() const;
  void items(std::vector<Item>* out) const;  // depth first, children in nibble order
  // create_trie_subset: nodes on the path of any key are kept, every other subtree becomes a hash node
  // (subtrees whose encoding is shorter than 32 bytes are embedded in their parent by the Yellow Paper's
  // rule and cannot be replaced by a hash without changing the root: they are kept).
  Status subset(const std::vector<Nibbles>& keys, Trie* out) const;
  // "a trie with just a hash node" (decoding.rs:474-478)
  Trie fully_hashed() const;

  // Byte form used across the C ABI: preorder, one tag byte per node.
  void serialize(Bytes* out) const;
  static Status deserialize(const uint8_t* d, size_t n, size_t* used, Trie* out);

 private:
  NodeP root_;
};

// every node encoding the trie's hasher runs Keccak-256 over (referenced by hash, or the root), children before parents
void hashed_node_preimages(const Trie& t, std::vector<Bytes>* out);
// hex-prefix encoding of a nibble path
Bytes hex_prefix(const Nibbles& k, bool leaf);
// node builders (shared with the compact-witness decoder, which builds tries structurally)
NodeP make_empty();
NodeP make_hash(const H256& h);
NodeP make_leaf(Nibbles key, Bytes value);
NodeP make_extension(Nibbles key, NodeP child);
NodeP make_branch(const NodeP (&children)[16], Bytes value = {});

}  // namespace mpt
