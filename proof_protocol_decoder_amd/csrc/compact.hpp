// compact.hpp -- what the compact-witness decoder (compact.cpp) hands to the IR producer (decoding.cpp):
// ProcessedCompactOutput of protocol_decoder/src/compact/compact_prestate_processing.rs:1243-1260.
#pragma once
#include <cstdlib>
#include <map>
#include <string>
#include <utility>
#include <vector>
#include "common.hpp"
#include "mpt.hpp"

namespace bpg {

struct CompactOut {
  uint8_t header_version = 0;
  mpt::Trie state;
  std::map<mpt::H256, mpt::Trie> storage_by_root;  // as extracted (keyed by storage root)
  std::map<mpt::H256, mpt::Trie> storage;          // keyed by hashed account address (compact_to_partial_trie.rs:167-190)
  std::map<mpt::H256, mpt::Bytes> code;            // code hash -> bytes
  std::vector<std::pair<mpt::H256, mpt::Account>> accounts;  // every account leaf of the state trie
};
bool decode_compact(const uint8_t* witness, size_t len, CompactOut* out, std::string* err);

// little helpers for the byte layouts that cross the C ABI
inline void put_u32(mpt::Bytes* o, uint32_t v) {
  for (int i = 0; i < 4; i++) o->push_back((uint8_t)(v >> (8 * i)));
}
inline void put_u64(mpt::Bytes* o, uint64_t v) {
  for (int i = 0; i < 8; i++) o->push_back((uint8_t)(v >> (8 * i)));
}
inline void put_blob(mpt::Bytes* o, const mpt::Bytes& b) {
  put_u32(o, (uint32_t)b.size());
  o->insert(o->end(), b.begin(), b.end());
}
inline mpt::Bytes trie_bytes(const mpt::Trie& t) {
  mpt::Bytes b;
  t.serialize(&b);
  return b;
}
inline int emit_bytes(const mpt::Bytes& o, uint8_t** out, size_t* out_len) {
  uint8_t* p = static_cast<uint8_t*>(std::malloc(o.size() ? o.size() : 1));
  if (!p) return fail(BP_ERR_DEVICE, "host allocation failed");
  if (!o.empty()) std::memcpy(p, o.data(), o.size());
  *out = p;
  *out_len = o.size();
  return BP_OK;
}

}  // namespace bpg
