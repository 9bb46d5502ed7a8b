// decoding.cpp -- SURVEY.md section 8(f) row 2: the txn IR producer.  A block trace (compact pre-image + per-txn
// account traces) becomes one self-contained GenerationInputs per transaction: minimal partial tries for what the
// txn touches, the txn's deltas replayed over the block's trie state, roots after, dummy padding, withdrawals.
//
// Restates, from the reference sources:
//   TxnInfo::into_processed_txn_info             protocol_decoder/src/processed_block_trace.rs:210-332
//   ProcessedBlockTrace::into_txn_proof_gen_ir   protocol_decoder/src/decoding.rs:81-177
//   create_minimal_partial_tries_needed_by_txn   decoding.rs:179-217 (+ :571-636)
//   apply_deltas_to_trie_state                   decoding.rs:219-292 (+ StateTrieWrites :431-456)
//   pad_gen_inputs_with_dummy_inputs_if_needed   decoding.rs:304-347 (+ dummy inputs :466-569)
//   add_withdrawals_to_txns / update_trie_state_from_withdrawals   decoding.rs:356-428
// Host C++ on purpose: each txn's tries depend on the previous txn's writes (inherently sequential) and the work
// is hash-map and trie bookkeeping ("not a resource bottleneck", README.md:9).  The trie library the reference
// uses (eth_trie_utils) is not in /root/reference; mpt.hpp restates what is needed of it.  The reference has no
// test for this path: tests/test_decoding.py pins it by invariants (replayed state root ==
// from-scratch recomputation, the asserts of decoding.rs:498-505, sub-tries hash to the full tries' roots).
//
// Two deliberate differences, both towards determinism: `traces` is a HashMap in the reference (iteration order
// unspecified, trace_protocol.rs:118); here accounts are processed in ascending address order.  And a dummy's
// "fully hashed out" tries are single hash nodes (the stated intent of decoding.rs:474-478) rather than
// create_trie_subset(trie, [0]).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>
#include "common.hpp"
#include "compact.hpp"

namespace {

using mpt::Bytes;
using mpt::H256;
using mpt::Nibbles;
using mpt::Trie;
using Addr = std::array<uint8_t, 20>;

// ---------------------------------------------------------------- input reader ("BPGTRAC1", include/bpg.h)
struct In {
  const uint8_t* p;
  size_t n, pos = 0;
  bool ok = true;
  bool need(size_t k) {
    if (!ok || n - pos < k) ok = false;
    return ok;
  }
  uint8_t u8() { return need(1) ? p[pos++] : 0; }
  uint32_t u32() {
    if (!need(4)) return 0;
    uint32_t v = (uint32_t)p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16) | ((uint32_t)p[pos + 3] << 24);
    pos += 4;
    return v;
  }
  uint64_t u64() {
    uint64_t lo = u32(), hi = u32();
    return lo | (hi << 32);
  }
  template <size_t N>
  std::array<uint8_t, N> fixed() {
    std::array<uint8_t, N> a{};
    if (need(N)) {
      std::memcpy(a.data(), p + pos, N);
      pos += N;
    }
    return a;
  }
  Bytes blob() {
    const uint32_t l = u32();
    Bytes b;
    if (need(l)) {
      b.assign(p + pos, p + pos + l);
      pos += l;
    }
    return b;
  }
};

// trace_protocol.rs:151-183
struct TxnTrace {
  Addr addr;
  bool has_balance = false, has_nonce = false, has_read = false, has_written = false, self_destructed = false;
  int code_kind = 0;  // 0 none, 1 read(hash), 2 write(bytes)
  H256 balance{}, nonce{}, code_hash{};
  Bytes code;
  std::vector<H256> storage_read;
  std::vector<std::pair<H256, H256>> storage_written;  // slot -> value (U256 big-endian)
};
struct TxnInfo {
  std::vector<TxnTrace> traces;
  Bytes byte_code, new_txn_trie_node_byte, new_receipt_trie_node_byte;
  uint64_t gas_used = 0;
};
struct OtherData {
  H256 checkpoint_state_trie_root{};
  Bytes block_metadata, block_hashes;  // upstream types, carried opaquely
  std::vector<std::pair<Addr, H256>> withdrawals;
  std::map<H256, Bytes> extra_code;    // CodeHashResolveFunc (types.rs:23) as a table
};

Bytes strip_be(const H256& v) {
  size_t i = 0;
  while (i < 32 && v[i] == 0) i++;
  return Bytes(v.begin() + i, v.end());
}
H256 pad_be(const Bytes& b) {
  H256 h{};
  std::memcpy(h.data() + 32 - b.size(), b.data(), b.size());
  return h;
}
H256 hash_of(const uint8_t* d, size_t n) { return mpt::keccak256(d, n); }

// processed_block_trace.rs:354-377
struct StateTrieWrites {
  bool has_balance = false, has_nonce = false, storage_trie_change = false, has_code_hash = false;
  H256 balance{}, nonce{}, code_hash{};
};
struct NodesUsedByTxn {
  std::vector<H256> state_accesses;
  std::vector<std::pair<H256, StateTrieWrites>> state_writes;
  std::vector<std::pair<H256, std::vector<Nibbles>>> storage_accesses;          // hashed account -> hashed slots
  std::vector<std::pair<H256, std::vector<std::pair<H256, Bytes>>>> storage_writes;  // hashed account -> (slot, rlp(value))
  std::map<H256, H256> accounts_with_no_accesses_but_storage_tries;
  std::vector<H256> self_destructed_accounts;
};
struct ProcessedTxnInfo {
  NodesUsedByTxn nodes;
  std::map<H256, Bytes> contract_code_accessed;
  bool has_txn_bytes = false;
  Bytes txn_bytes, receipt_node_bytes;
  uint64_t gas_used = 0;
};

// process_rlped_receipt_node_bytes (processed_block_trace.rs:335-343): a legacy receipt
// rlp([status, cum_gas_used, bloom(256 bytes), logs]) is kept as is; anything else must be an RLP string
// wrapping a typed receipt and is unwrapped.
bool receipt_node_bytes(const Bytes& raw, Bytes* out, std::string* err) {
  mpt::RlpItem top;
  if (mpt::rlp_parse(raw.data(), raw.size(), &top) && top.total == raw.size()) {
    if (top.is_list) {
      std::vector<mpt::RlpItem> f;
      if (mpt::rlp_children(top, &f) && f.size() == 4 && !f[0].is_list && !f[1].is_list && !f[2].is_list &&
          f[2].len == 256 && f[3].is_list) {
        *out = raw;
        return true;
      }
    } else {
      out->assign(top.payload, top.payload + top.len);
      return true;
    }
  }
  *err = "new_receipt_trie_node_byte is neither a legacy receipt nor an RLP byte string";
  return false;
}

// TxnInfo::into_processed_txn_info (processed_block_trace.rs:210-332)
bool process_txn_info(const TxnInfo& t, const std::vector<std::pair<H256, mpt::Account>>& all_accounts,
                      const std::map<H256, Bytes>& witness_code, const OtherData& od, ProcessedTxnInfo* out,
                      std::string* err) {
  NodesUsedByTxn& n = out->nodes;
  out->contract_code_accessed[mpt::EMPTY_CODE_HASH] = Bytes{};  // create_empty_code_access_map
  std::vector<const TxnTrace*> order;
  for (auto& tr : t.traces) order.push_back(&tr);
  std::sort(order.begin(), order.end(), [](const TxnTrace* a, const TxnTrace* b) { return a->addr < b->addr; });
  for (const TxnTrace* trp : order) {
    const TxnTrace& tr = *trp;
    const H256 hashed_addr = hash_of(tr.addr.data(), 20);
    std::vector<Nibbles> access;  // reads, then written keys
    for (auto& k : tr.storage_read) access.push_back(mpt::nibbles_of(hash_of(k.data(), 32)));
    for (auto& kv : tr.storage_written) access.push_back(mpt::nibbles_of(hash_of(kv.first.data(), 32)));
    n.storage_accesses.push_back({hashed_addr, access});
    const bool storage_trie_change = !tr.storage_written.empty();
    const bool code_change = tr.code_kind != 0;
    if (tr.has_balance || tr.has_nonce || storage_trie_change || code_change) {
      StateTrieWrites w;
      w.has_balance = tr.has_balance; w.balance = tr.balance;
      w.has_nonce = tr.has_nonce; w.nonce = tr.nonce;
      w.storage_trie_change = storage_trie_change;
      if (tr.code_kind == 1) { w.has_code_hash = true; w.code_hash = tr.code_hash; }
      if (tr.code_kind == 2) { w.has_code_hash = true; w.code_hash = mpt::keccak256(tr.code); }
      n.state_writes.push_back({hashed_addr, w});
    }
    std::vector<std::pair<H256, Bytes>> writes;
    for (auto& kv : tr.storage_written) writes.push_back({kv.first, mpt::rlp_scalar_be(Bytes(kv.second.begin(), kv.second.end()))});
    n.storage_writes.push_back({hashed_addr, writes});
    n.state_accesses.push_back(hashed_addr);
    if (tr.code_kind == 1 && !out->contract_code_accessed.count(tr.code_hash)) {
      // extra_code_hash_mappings of the witness first, then the caller's resolver (:70-82)
      auto w = witness_code.find(tr.code_hash);
      auto e = od.extra_code.find(tr.code_hash);
      if (w != witness_code.end()) out->contract_code_accessed[tr.code_hash] = w->second;
      else if (e != od.extra_code.end()) out->contract_code_accessed[tr.code_hash] = e->second;
      else {
        *err = "code hash read by a transaction is neither in the witness nor in the caller's code table";
        return false;
      }
    }
    if (tr.code_kind == 2) out->contract_code_accessed[mpt::keccak256(tr.code)] = tr.code;
    if (tr.self_destructed) n.self_destructed_accounts.push_back(hashed_addr);
  }
  std::set<H256> with_accesses;
  for (auto& sa : n.storage_accesses)
    if (!sa.second.empty()) with_accesses.insert(sa.first);
  for (auto& acc : all_accounts)
    if (acc.second.storage_root != mpt::EMPTY_TRIE_HASH && !with_accesses.count(acc.first))
      n.accounts_with_no_accesses_but_storage_tries[acc.first] = acc.second.storage_root;
  out->has_txn_bytes = !t.byte_code.empty();
  out->txn_bytes = t.byte_code;
  out->gas_used = t.gas_used;
  return receipt_node_bytes(t.new_receipt_trie_node_byte, &out->receipt_node_bytes, err);
}

// decoding.rs:72-79
struct PartialTrieState {
  Trie state, txn, receipt;
  std::map<H256, Trie> storage;
};
struct TrieInputs {
  Trie state, transactions, receipts;
  std::vector<std::pair<H256, Trie>> storage;
};
struct GenInputs {
  H256 txn_number_before{}, gas_used_before{}, gas_used_after{};  // U256 big-endian
  bool has_signed_txn = false;
  Bytes signed_txn;
  std::vector<std::pair<Addr, H256>> withdrawals;
  TrieInputs tries;
  H256 root_state{}, root_txn{}, root_receipt{};
  std::map<H256, Bytes> contract_code;
};

H256 u256_of(uint64_t v) {
  H256 h{};
  for (int i = 0; i < 8; i++) h[31 - i] = (uint8_t)(v >> (8 * i));
  return h;
}
H256 u256_add(const H256& a, uint64_t b) {
  bool ovf = false;
  return pad_be(mpt::u256_add_be(strip_be(a), strip_be(u256_of(b)), &ovf));
}
Nibbles txn_key(size_t idx) {  // Nibbles::from_bytes_be(&rlp::encode(&txn_idx)) (decoding.rs:190, 283)
  const Bytes r = mpt::rlp_u64(idx);
  return mpt::nibbles_of_bytes(r.data(), r.size());
}
int trie_fail(mpt::Status s, const char* what) {
  return bpg::fail(BP_ERR_INVALID_INPUT, "%s: %s", what, mpt::status_text(s));
}

// create_minimal_partial_tries_needed_by_txn (decoding.rs:179-217, 571-636)
int minimal_tries(PartialTrieState& cur, const NodesUsedByTxn& n, size_t txn_idx, TrieInputs* out) {
  std::vector<Nibbles> keys;
  for (auto& h : n.state_accesses) keys.push_back(mpt::nibbles_of(h));
  mpt::Status s = cur.state.subset(keys, &out->state);
  if (s != mpt::Status::Ok) return trie_fail(s, "Missing keys when creating sub-partial tries (Trie type: state)");
  const std::vector<Nibbles> tk{txn_key(txn_idx)};
  if ((s = cur.txn.subset(tk, &out->transactions)) != mpt::Status::Ok)
    return trie_fail(s, "Missing keys when creating sub-partial tries (Trie type: transaction)");
  if ((s = cur.receipt.subset(tk, &out->receipts)) != mpt::Status::Ok)
    return trie_fail(s, "Missing keys when creating sub-partial tries (Trie type: receipt)");
  for (auto& sa : n.storage_accesses) {
    auto it = cur.storage.find(sa.first);
    if (it == cur.storage.end()) {
      // the reference's "big hack" (:593-607): an account without a storage trie in the pre-image gets a
      // hash node of its storage root if it is known, else an empty trie, and the block state keeps it
      auto known = n.accounts_with_no_accesses_but_storage_tries.find(sa.first);
      it = cur.storage.insert({sa.first, known != n.accounts_with_no_accesses_but_storage_tries.end()
                                             ? Trie::of_hash(known->second) : Trie()}).first;
    }
    Trie sub;
    if ((s = it->second.subset(sa.second, &sub)) != mpt::Status::Ok)
      return trie_fail(s, "Missing keys when creating sub-partial tries (Trie type: storage)");
    out->storage.push_back({sa.first, sub});
  }
  return BP_OK;
}

int account_of(const Bytes& rlp, mpt::Account* a) {
  if (!mpt::account_decode(rlp, a))
    return bpg::fail(BP_ERR_INVALID_INPUT, "Failed to decode RLP bytes as an Ethereum account");
  return BP_OK;
}
const Bytes& empty_account_rlp() {  // EMPTY_ACCOUNT_BYTES_RLPED (types.rs:36-43)
  static const Bytes b = mpt::account_encode(mpt::Account{{}, {}, mpt::EMPTY_TRIE_HASH, mpt::EMPTY_CODE_HASH});
  return b;
}

// apply_deltas_to_trie_state (decoding.rs:219-292)
int apply_deltas(PartialTrieState& st, const ProcessedTxnInfo& t, size_t txn_idx) {
  const NodesUsedByTxn& d = t.nodes;
  for (auto& sw : d.storage_writes) {
    auto it = st.storage.find(sw.first);
    if (it == st.storage.end()) {
      if (sw.second.empty()) continue;  // the reference looks the trie up before looping; nothing to write is harmless
      return bpg::fail(BP_ERR_INVALID_INPUT, "Missing account storage trie in base trie when applying storage writes");
    }
    for (auto& kv : sw.second) {
      const Nibbles slot = mpt::nibbles_of(hash_of(kv.first.data(), 32));
      mpt::Status s;
      if (kv.second == Bytes{0x80}) {  // writing rlp(0) is a delete (ZERO_STORAGE_SLOT_VAL_RLPED)
        bool existed;
        s = it->second.remove(slot, &existed);
      } else {
        s = it->second.insert(slot, kv.second);
      }
      if (s != mpt::Status::Ok) return trie_fail(s, "storage write");
    }
  }
  for (auto& sw : d.state_writes) {
    const Nibbles k = mpt::nibbles_of(sw.first);
    const Bytes* cur = nullptr;
    mpt::Status s = st.state.get(k, &cur);
    if (s != mpt::Status::Ok) return trie_fail(s, "state write");
    mpt::Account acc;
    int rc = account_of(cur ? *cur : empty_account_rlp(), &acc);  // a created account is not in the trie yet
    if (rc) return rc;
    const StateTrieWrites& w = sw.second;  // StateTrieWrites::apply_writes_to_state_node (:431-456)
    if (w.has_balance) acc.balance_be = strip_be(w.balance);
    if (w.has_nonce) acc.nonce_be = strip_be(w.nonce);
    if (w.storage_trie_change) {
      auto it = st.storage.find(sw.first);
      if (it == st.storage.end()) return bpg::fail(BP_ERR_INVALID_INPUT, "Missing account storage trie in base trie (state write)");
      acc.storage_root = it->second.hash();
    }
    if (w.has_code_hash) acc.code_hash = w.code_hash;
    if ((s = st.state.insert(k, mpt::account_encode(acc))) != mpt::Status::Ok) return trie_fail(s, "state write");
  }
  for (auto& h : d.self_destructed_accounts) {
    if (!st.storage.erase(h)) return bpg::fail(BP_ERR_INVALID_INPUT, "Missing account storage trie of a self-destructed account");
    bool existed;
    mpt::Status s = st.state.remove(mpt::nibbles_of(h), &existed);
    if (s != mpt::Status::Ok) return trie_fail(s, "self-destruct");
  }
  const Nibbles tk = txn_key(txn_idx);
  mpt::Status s = st.txn.insert(tk, t.has_txn_bytes ? t.txn_bytes : Bytes{});
  if (s == mpt::Status::Ok) s = st.receipt.insert(tk, t.receipt_node_bytes);
  if (s != mpt::Status::Ok) return trie_fail(s, "txn / receipt trie insert");
  return BP_OK;
}

// create_dummy_gen_input (decoding.rs:484-520)
GenInputs dummy_input(const PartialTrieState& tries, const H256& txn_number, const H256& gas_used) {
  GenInputs g;
  g.tries.state = tries.state.fully_hashed();
  g.tries.transactions = tries.txn.fully_hashed();
  g.tries.receipts = tries.receipt.fully_hashed();
  for (auto& kv : tries.storage) g.tries.storage.push_back({kv.first, kv.second.fully_hashed()});
  g.root_state = g.tries.state.hash();
  g.root_txn = g.tries.transactions.hash();
  g.root_receipt = g.tries.receipts.hash();
  g.txn_number_before = txn_number;  // the asserts of :498-505 hold by construction: before == after
  g.gas_used_before = g.gas_used_after = gas_used;
  return g;
}

// update_trie_state_from_withdrawals (decoding.rs:404-428)
int apply_withdrawals(const std::vector<std::pair<Addr, H256>>& w, Trie* state) {
  for (auto& wd : w) {
    const Nibbles k = mpt::nibbles_of(hash_of(wd.first.data(), 20));
    const Bytes* cur = nullptr;
    mpt::Status s = state->get(k, &cur);
    if (s != mpt::Status::Ok) return trie_fail(s, "withdrawal");
    if (!cur) return bpg::fail(BP_ERR_INVALID_INPUT, "No account present to withdraw to");
    mpt::Account acc;
    int rc = account_of(*cur, &acc);
    if (rc) return rc;
    bool ovf = false;
    acc.balance_be = mpt::u256_add_be(acc.balance_be, strip_be(wd.second), &ovf);
    if (ovf) return bpg::fail(BP_ERR_INVALID_INPUT, "withdrawal overflows the account balance");
    if ((s = state->insert(k, mpt::account_encode(acc))) != mpt::Status::Ok) return trie_fail(s, "withdrawal");
  }
  return BP_OK;
}

void put_fixed(Bytes* o, const uint8_t* d, size_t n) { o->insert(o->end(), d, d + n); }
void put_gen_inputs(Bytes* o, const GenInputs& g, const OtherData& od) {
  put_fixed(o, g.txn_number_before.data(), 32);
  put_fixed(o, g.gas_used_before.data(), 32);
  put_fixed(o, g.gas_used_after.data(), 32);
  o->push_back(g.has_signed_txn ? 1 : 0);
  bpg::put_blob(o, g.signed_txn);
  bpg::put_u32(o, (uint32_t)g.withdrawals.size());
  for (auto& w : g.withdrawals) {
    put_fixed(o, w.first.data(), 20);
    put_fixed(o, w.second.data(), 32);
  }
  bpg::put_blob(o, bpg::trie_bytes(g.tries.state));
  bpg::put_blob(o, bpg::trie_bytes(g.tries.transactions));
  bpg::put_blob(o, bpg::trie_bytes(g.tries.receipts));
  bpg::put_u32(o, (uint32_t)g.tries.storage.size());
  for (auto& s : g.tries.storage) {
    put_fixed(o, s.first.data(), 32);
    bpg::put_blob(o, bpg::trie_bytes(s.second));
  }
  put_fixed(o, g.root_state.data(), 32);
  put_fixed(o, g.root_txn.data(), 32);
  put_fixed(o, g.root_receipt.data(), 32);
  put_fixed(o, od.checkpoint_state_trie_root.data(), 32);
  bpg::put_u32(o, (uint32_t)g.contract_code.size());
  for (auto& c : g.contract_code) {
    put_fixed(o, c.first.data(), 32);
    bpg::put_blob(o, c.second);
  }
  bpg::put_blob(o, od.block_metadata);
  bpg::put_blob(o, od.block_hashes);
}

bool read_trace(In& in, TxnTrace* t) {
  t->addr = in.fixed<20>();
  const uint8_t f = in.u8();
  if (f & 0x80 || ((f & 16) && (f & 32))) in.ok = false;
  if (f & 1) { t->has_balance = true; t->balance = in.fixed<32>(); }
  if (f & 2) { t->has_nonce = true; t->nonce = in.fixed<32>(); }
  if (f & 4) {
    t->has_read = true;
    const uint32_t n = in.u32();
    for (uint32_t i = 0; i < n && in.ok; i++) t->storage_read.push_back(in.fixed<32>());
  }
  if (f & 8) {
    t->has_written = true;
    const uint32_t n = in.u32();
    for (uint32_t i = 0; i < n && in.ok; i++) {
      const H256 k = in.fixed<32>(), v = in.fixed<32>();
      t->storage_written.push_back({k, v});
    }
  }
  if (f & 16) { t->code_kind = 1; t->code_hash = in.fixed<32>(); }
  if (f & 32) { t->code_kind = 2; t->code = in.blob(); }
  t->self_destructed = f & 64;
  return in.ok;
}

}  // namespace

extern "C" {

// BlockTrace::into_txn_proof_gen_ir (processed_block_trace.rs:38-50 -> decoding.rs:81-177).
int bp_decode_block_trace(const uint8_t* trace, size_t len, uint8_t** out, size_t* out_len) try {
  if (!trace || !out || !out_len) return bpg::fail(BP_ERR_INVALID_INPUT, "bp_decode_block_trace: null argument");
  In in{trace, len};
  if (len < 8 || std::memcmp(trace, "BPGTRAC1", 8) != 0) return bpg::fail(BP_ERR_INVALID_INPUT, "block trace: bad magic");
  in.pos = 8;
  const Bytes witness = in.blob();
  // a transaction is at least 24 bytes of this form: a count larger than the bytes left is malformed, not a
  // reason to allocate
  const uint32_t n_txn = in.u32();
  if (!in.ok || n_txn > (len - in.pos) / 24) return bpg::fail(BP_ERR_INVALID_INPUT, "block trace: transaction count exceeds the payload");
  std::vector<TxnInfo> txns(n_txn);
  for (auto& t : txns) {
    const uint32_t nt = in.u32();
    for (uint32_t i = 0; i < nt && in.ok; i++) {
      t.traces.emplace_back();
      read_trace(in, &t.traces.back());
    }
    t.byte_code = in.blob();
    t.new_txn_trie_node_byte = in.blob();
    t.new_receipt_trie_node_byte = in.blob();
    t.gas_used = in.u64();
    if (!in.ok) break;
  }
  OtherData od;
  od.checkpoint_state_trie_root = in.fixed<32>();
  od.block_metadata = in.blob();
  od.block_hashes = in.blob();
  for (uint32_t i = 0, n = in.u32(); i < n && in.ok; i++) {
    const Addr a = in.fixed<20>();
    const H256 amt = in.fixed<32>();
    od.withdrawals.push_back({a, amt});
  }
  for (uint32_t i = 0, n = in.u32(); i < n && in.ok; i++) {
    const H256 h = in.fixed<32>();
    od.extra_code[h] = in.blob();
  }
  if (!in.ok || in.pos != len) return bpg::fail(BP_ERR_INVALID_INPUT, "block trace: truncated or trailing bytes (offset %zu of %zu)", in.pos, len);

  // process_compact_trie (processed_block_trace.rs:170-181)
  bpg::CompactOut pre;
  std::string err;
  if (!bpg::decode_compact(witness.data(), witness.size(), &pre, &err)) return bpg::fail(BP_ERR_INVALID_INPUT, "%s", err.c_str());
  if (pre.header_version != 1)
    return bpg::fail(BP_ERR_INVALID_INPUT, "compact witness header version %u is not the supported version 1", pre.header_version);

  std::vector<ProcessedTxnInfo> infos(txns.size());
  for (size_t i = 0; i < txns.size(); i++)
    if (!process_txn_info(txns[i], pre.accounts, pre.code, od, &infos[i], &err)) return bpg::fail(BP_ERR_INVALID_INPUT, "txn %zu: %s", i, err.c_str());

  PartialTrieState cur;
  cur.state = pre.state;
  cur.storage = pre.storage;
  const PartialTrieState initial = cur;  // initial_tries_for_dummies: shares every node with `cur`
  H256 txn_number{}, gas_used{};
  std::vector<GenInputs> irs;
  for (size_t i = 0; i < infos.size(); i++) {
    GenInputs g;
    int rc = minimal_tries(cur, infos[i].nodes, i, &g.tries);
    if (rc) return rc;
    g.txn_number_before = txn_number;
    g.gas_used_before = gas_used;
    gas_used = u256_add(gas_used, infos[i].gas_used);
    txn_number = u256_add(txn_number, 1);
    g.gas_used_after = gas_used;
    if ((rc = apply_deltas(cur, infos[i], i))) return rc;
    g.root_state = cur.state.hash();
    g.root_txn = cur.txn.hash();
    g.root_receipt = cur.receipt.hash();
    g.has_signed_txn = infos[i].has_txn_bytes;
    g.signed_txn = infos[i].txn_bytes;
    g.contract_code = infos[i].contract_code_accessed;
    irs.push_back(std::move(g));
  }
  // pad_gen_inputs_with_dummy_inputs_if_needed (decoding.rs:304-347)
  const bool has_withdrawals = !od.withdrawals.empty();
  bool dummies_added = false;
  if (irs.empty()) {
    irs.push_back(dummy_input(initial, txn_number, gas_used));
    irs.push_back(dummy_input(initial, txn_number, gas_used));
    dummies_added = true;
  } else if (irs.size() == 1) {
    // the reference builds this dummy from extra_data AFTER the only txn (txn number 1, the block's gas) in
    // both placements (:329-338); kept as is
    if (has_withdrawals) irs.push_back(dummy_input(cur, txn_number, gas_used));
    else irs.insert(irs.begin(), dummy_input(initial, txn_number, gas_used));
    dummies_added = true;
  }
  // add_withdrawals_to_txns (decoding.rs:356-402)
  if (has_withdrawals) {
    if (!dummies_added) {
      GenInputs wd = dummy_input(cur, txn_number, gas_used);
      int rc = apply_withdrawals(od.withdrawals, &cur.state);
      if (rc) return rc;
      wd.withdrawals = od.withdrawals;
      wd.root_state = cur.state.hash();
      irs.push_back(std::move(wd));
    } else {
      int rc = apply_withdrawals(od.withdrawals, &cur.state);
      if (rc) return rc;
      irs[1].withdrawals = od.withdrawals;  // irs[1] is always a dummy here
      irs[1].root_state = cur.state.hash();
    }
  }
  Bytes o;
  o.insert(o.end(), {'B', 'P', 'G', 'G', 'E', 'N', 'I', '1'});
  bpg::put_u32(&o, (uint32_t)irs.size());
  for (auto& g : irs) put_gen_inputs(&o, g, od);
  // the block's final trie state, for the caller's checks: state root and every storage trie root
  const H256 fin = cur.state.hash();
  put_fixed(&o, fin.data(), 32);
  return bpg::emit_bytes(o, out, out_len);
}
BPG_ABI_CATCH("bp_decode_block_trace")

}  // extern "C"
