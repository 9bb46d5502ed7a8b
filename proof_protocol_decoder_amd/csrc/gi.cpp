// gi.cpp -- the two pieces of host logic that sit between the decoder and the prover in the reference's callers, on
// the library side of the C ABI (they lived in Python, block_driver.py, where a Rust host cannot
// link them):
//
//  * GenerationInputs as the prover's input.  `generate_txn_proof(&ProverState, TxnProofGenIR = GenerationInputs, abort)`
//    (plonky_block_proof_gen/src/proof_gen.rs:39-43, protocol_decoder/src/types.rs:48; the fields as populated at
//    protocol_decoder/src/decoding.rs:131-145) takes one entry of what the decoder emits.  bp_generate_txn_proof_gi
//    takes one entry of a "BPGGENI1" buffer (bp_decode_block_trace) and derives from it everything the synthetic
//    prover consumes: the 25-word IR (counters and the state-root chain threaded entry to entry as decoding.rs:106-154
//    threads them; the witness seed = Keccak-256 of the entry's own data) and, for the tables proven with their AIRs,
//    the witness of the entry's OWN hashing work -- Keccak-f permutations, sponge rows, the memory log and the
//    byte-packing sequences of the same bytes.
//  * The shard scheduler (the reference leaves scheduling to "Paladin", docs/usage_seq_diagrams.md:8-20): all txn
//    proofs of a contiguous slice and its aggregation tree on a pool of threads, every aggregation going AHEAD of the
//    transactions still waiting for a thread.  The headline rate of bench.py is a property of this policy as much as of
//    the kernels, so it belongs to the library.
//
// Built on the public ABI (include/bpg.h) only -- what a host in any language could do itself.
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <queue>
#include <string>
#include <system_error>
#include <thread>
#include <vector>
#include "common.hpp"
#include "gl.hpp"
#include "mpt.hpp"

namespace {

using bpg::fail;
using mpt::Bytes;
using mpt::H256;

// ---------------------------------------------------------------- "BPGGENI1" reader (layout: include/bpg.h)
struct View {
  const uint8_t* p = nullptr;
  size_t n = 0;
};
struct Rd {
  const uint8_t* d;
  size_t n, pos = 0;
  bool ok = true;
  const uint8_t* take(size_t k) {
    if (!ok || k > n - pos) { ok = false; return nullptr; }
    const uint8_t* r = d + pos;
    pos += k;
    return r;
  }
  uint8_t u8() { const uint8_t* r = take(1); return r ? *r : 0; }
  uint32_t u32() {
    const uint8_t* r = take(4);
    return r ? (uint32_t)r[0] | (uint32_t)r[1] << 8 | (uint32_t)r[2] << 16 | (uint32_t)r[3] << 24 : 0;
  }
  View blob() {
    const uint32_t k = u32();
    const uint8_t* r = take(k);
    return View{r, r ? k : 0};
  }
};
struct Entry {
  const uint8_t *txn_number_before, *gas_used_before, *gas_used_after;  // U256, 32 bytes big-endian
  bool has_signed_txn;
  View signed_txn;
  std::vector<std::pair<const uint8_t*, const uint8_t*>> withdrawals;  // address[20], amount[32]
  View tries[3];                                                       // state, transactions, receipts
  std::vector<std::pair<const uint8_t*, View>> storage;                // hashed address[32], trie
  const uint8_t* roots_after[3];                                       // state, transactions, receipts
  const uint8_t* checkpoint;
  std::vector<std::pair<const uint8_t*, View>> code;                   // code hash[32], bytes (ascending hash order)
  View block_metadata, block_hashes;
};
bool read_entry(Rd& r, Entry* e) {
  e->txn_number_before = r.take(32);
  e->gas_used_before = r.take(32);
  e->gas_used_after = r.take(32);
  e->has_signed_txn = r.u8() != 0;
  e->signed_txn = r.blob();
  for (uint32_t i = 0, n = r.u32(); i < n && r.ok; i++) {
    const uint8_t* a = r.take(20);
    const uint8_t* v = r.take(32);
    e->withdrawals.push_back({a, v});
  }
  for (int t = 0; t < 3; t++) e->tries[t] = r.blob();
  for (uint32_t i = 0, n = r.u32(); i < n && r.ok; i++) {
    const uint8_t* h = r.take(32);
    e->storage.push_back({h, r.blob()});
  }
  for (int t = 0; t < 3; t++) e->roots_after[t] = r.take(32);
  e->checkpoint = r.take(32);
  for (uint32_t i = 0, n = r.u32(); i < n && r.ok; i++) {
    const uint8_t* h = r.take(32);
    e->code.push_back({h, r.blob()});
  }
  e->block_metadata = r.blob();
  e->block_hashes = r.blob();
  return r.ok;
}
int parse_geni(const uint8_t* geni, size_t len, std::vector<Entry>* out) {
  if (!geni || len < 12 || std::memcmp(geni, "BPGGENI1", 8) != 0) return fail(BP_ERR_INVALID_INPUT, "generation inputs: bad magic (expected \"BPGGENI1\", the output of bp_decode_block_trace)");
  Rd r{geni, len};
  r.pos = 8;
  const uint32_t n = r.u32();
  if (n > (len - r.pos) / 200) return fail(BP_ERR_INVALID_INPUT, "generation inputs: entry count exceeds the buffer");
  out->resize(n);
  for (uint32_t i = 0; i < n; i++)
    if (!read_entry(r, &(*out)[i])) return fail(BP_ERR_INVALID_INPUT, "generation inputs: truncated at entry %u", i);
  r.take(32);  // the block's final state root
  if (!r.ok || r.pos != len) return fail(BP_ERR_INVALID_INPUT, "generation inputs: truncated or trailing bytes");
  return BP_OK;
}
int u256_low64(const uint8_t* be, uint64_t* out, const char* what) {
  for (int i = 0; i < 24; i++)
    if (be[i]) return fail(BP_ERR_INVALID_INPUT, "generation inputs: %s does not fit 64 bits", what);
  uint64_t v = 0;
  for (int i = 24; i < 32; i++) v = v << 8 | be[i];
  *out = v;
  return BP_OK;
}
uint64_t le64(const uint8_t* p) {
  uint64_t v = 0;
  for (int i = 7; i >= 0; i--) v = v << 8 | p[i];
  return v;
}
int trie_of(const View& v, mpt::Trie* out, const char* what) {
  size_t used = 0;
  if (mpt::Trie::deserialize(v.p, v.n, &used, out) != mpt::Status::Ok || used != v.n)
    return fail(BP_ERR_INVALID_INPUT, "generation inputs: malformed %s trie", what);
  return BP_OK;
}

// ---------------------------------------------------------------- the entry's own hashing work
// The byte strings a zkEVM proving this entry hashes before it executes anything: signed_txn (the transaction hash), every
// contract_code entry (the code hashes the decoder keys them by, decoding.rs:131-145; ascending hash order) and, with
// BP_GI_KECCAK_TRIE_NODES, every hash-referenced node of the entry's partial tries (decoding.rs:179-217: state,
// transactions, receipts, storage tries in their order; children before parents).  What remains upstream-only is the
// hashing done WHILE executing.
int hashed_preimages(const Entry& e, bool trie_nodes, std::vector<Bytes>* out) {
  if (e.signed_txn.n) out->emplace_back(e.signed_txn.p, e.signed_txn.p + e.signed_txn.n);
  std::map<H256, View> by_hash;
  for (auto& c : e.code) {
    H256 h;
    std::memcpy(h.data(), c.first, 32);
    by_hash[h] = c.second;
  }
  for (auto& c : by_hash) {
    if (mpt::keccak256(c.second.p, c.second.n) != c.first)
      return fail(BP_ERR_INVALID_INPUT, "generation inputs: contract_code is keyed by a hash that is not the Keccak-256 of its bytes");
    out->emplace_back(c.second.p, c.second.p + c.second.n);
  }
  if (trie_nodes) {
    std::vector<View> tries = {e.tries[0], e.tries[1], e.tries[2]};
    for (auto& s : e.storage) tries.push_back(s.second);
    for (const View& v : tries) {
      mpt::Trie t;
      if (int rc = trie_of(v, &t, "partial")) return rc;
      const size_t before = out->size();
      mpt::hashed_node_preimages(t, out);
      if (out->size() > before && mpt::keccak256(out->back()) != t.hash())
        return fail(BP_ERR_INVALID_INPUT, "generation inputs: a partial trie's node hashes do not end in its root");
    }
  }
  return BP_OK;
}
// What moving those strings to the hasher looks like (AIRS.md section 3, lookup byte_packing -> memory): 32 bytes at a
// time, one byte-packing sequence per chunk -- [is_read = 1 | timestamp << 8, len | address << 8, four words of byte
// slots] -- and the 256-bit word a chunk spells lives at its own address, written once and read once by the packer:
// memory log [is_read, address, timestamp, eight 32-bit limbs], sorted by (address, timestamp).
void memory_and_byte_packing_work(const std::vector<Bytes>& pre, std::vector<uint64_t>* log, std::vector<uint64_t>* seqs) {
  uint64_t addr = 0;
  for (const Bytes& m : pre)
    for (size_t off = 0; off < m.size(); off += 32) {
      const size_t k = std::min<size_t>(32, m.size() - off);
      uint8_t be[32] = {0}, padded[32] = {0};
      std::memcpy(be + 32 - k, m.data() + off, k);   // the chunk as a big-endian integer
      std::memcpy(padded, m.data() + off, k);
      uint64_t limbs[8];
      for (int j = 0; j < 8; j++) {
        const uint8_t* q = be + 28 - 4 * j;
        limbs[j] = (uint64_t)q[0] << 24 | (uint64_t)q[1] << 16 | (uint64_t)q[2] << 8 | q[3];
      }
      const uint64_t ts = addr + 2;
      const uint64_t wr[3] = {0, addr, 1}, rd[3] = {1, addr, ts};
      log->insert(log->end(), wr, wr + 3);
      log->insert(log->end(), limbs, limbs + 8);
      log->insert(log->end(), rd, rd + 3);
      log->insert(log->end(), limbs, limbs + 8);
      seqs->push_back(1 | ts << 8);
      seqs->push_back(k | addr << 8);
      for (int w = 0; w < 4; w++) seqs->push_back(le64(padded + 8 * w));
      addr++;
    }
}
uint32_t ceil_log2(uint64_t n) {
  uint32_t l = 0;
  while (((uint64_t)1 << l) < n) l++;
  return l;
}

struct EntryWork {  // everything bp_generate_txn_proof_witness needs for one entry
  uint64_t ir[BP_IR_WORDS];
  bool any_witness = false;
  std::vector<uint64_t> perms, rows, log, seqs;
};
// IR (+ witness when `with_witness`) of entry e given the chain before it; *chain moves to after the entry.
int entry_work(const Entry& e, const bp_gi_options& o, bp_gi_chain* chain, bool with_witness, EntryWork* w) {
  const uint32_t f = o.flags;
  if (f & ~(uint32_t)63) return fail(BP_ERR_INVALID_INPUT, "bp_gi_options.flags: unknown bits");
  if ((f & BP_GI_LOGIC_AIR) && !(f & BP_GI_KECCAK_SPONGE_AIR))
    return fail(BP_ERR_INVALID_INPUT, "the logic table's work is the sponge table's XORs: BP_GI_LOGIC_AIR needs BP_GI_KECCAK_SPONGE_AIR");
  if ((f & (BP_GI_MEMORY_AIR | BP_GI_BYTE_PACKING_AIR | BP_GI_KECCAK_SPONGE_AIR | BP_GI_KECCAK_TRIE_NODES)) && !(f & BP_GI_KECCAK_AIR))
    return fail(BP_ERR_INVALID_INPUT, "the memory / byte-packing / sponge work is that of the hashed bytes: it needs BP_GI_KECCAK_AIR");
  uint32_t log_n[BP_NUM_TABLES], width[BP_NUM_TABLES];
  std::memcpy(log_n, o.table_log_n, sizeof(log_n));
  std::memcpy(width, o.table_width, sizeof(width));
  if (f & BP_GI_KECCAK_AIR) width[3] = 2431;
  if (f & BP_GI_KECCAK_SPONGE_AIR) width[4] = 2414;
  if (f & BP_GI_MEMORY_AIR) width[6] = 45;
  if (f & BP_GI_BYTE_PACKING_AIR) width[1] = 299;
  if (f & BP_GI_LOGIC_AIR) width[5] = 524;
  if (f & BP_GI_KECCAK_AIR) {
    // the heights grow to hold the work (24 rows per permutation); the witness itself only when it is asked for
    std::vector<Bytes> pre;
    if (int rc = hashed_preimages(e, (f & BP_GI_KECCAK_TRIE_NODES) != 0, &pre)) return rc;
    size_t n_perms = 0, n_rows = 0, n_chunks = 0;
    for (const Bytes& m : pre) {
      n_perms += m.size() / 136 + 1;
      n_rows += m.size() / 136 + 1;
      n_chunks += (m.size() + 31) / 32;
    }
    log_n[3] = std::max(log_n[3], ceil_log2(std::max<uint64_t>(24 * n_perms, 1)));
    if (f & BP_GI_KECCAK_SPONGE_AIR) log_n[4] = std::max(log_n[4], ceil_log2(std::max<uint64_t>(n_rows, 1)));
    // (five XORs per absorbed block, of every row of the sponge table's final height: keccak_sponge -> logic)
    if (f & BP_GI_LOGIC_AIR) log_n[5] = std::max(log_n[5], ceil_log2(5ull << log_n[4]));
    if (f & BP_GI_MEMORY_AIR) log_n[6] = std::max(log_n[6], ceil_log2(std::max<uint64_t>(2 * n_chunks, 1)));
    if (f & BP_GI_BYTE_PACKING_AIR) log_n[1] = std::max(log_n[1], ceil_log2(std::max<uint64_t>(n_chunks, 1)));
    if (with_witness) {
      w->any_witness = true;
      for (const Bytes& m : pre) {
        mpt::keccak256_traced(m.data(), m.size(), &w->perms);
        if (f & BP_GI_KECCAK_SPONGE_AIR) mpt::keccak256_sponge_rows(m.data(), m.size(), &w->rows);
      }
      if (f & (BP_GI_MEMORY_AIR | BP_GI_BYTE_PACKING_AIR)) memory_and_byte_packing_work(pre, &w->log, &w->seqs);
    }
  }
  // the witness seed binds the proof to the decoded state transition: keccak(signed_txn | roots after | withdrawals)
  Bytes blob;
  if (e.signed_txn.n) blob.insert(blob.end(), e.signed_txn.p, e.signed_txn.p + e.signed_txn.n);
  for (int t = 0; t < 3; t++) blob.insert(blob.end(), e.roots_after[t], e.roots_after[t] + 32);
  for (auto& wd : e.withdrawals) {
    blob.insert(blob.end(), wd.first, wd.first + 20);
    blob.insert(blob.end(), wd.second, wd.second + 32);
  }
  const H256 digest = mpt::keccak256(blob);
  const uint64_t seed = le64(digest.data());
  int rc;
  if (!e.has_signed_txn) {
    // an entry without a transaction (dummy padding, the withdrawal carrier): proven, the counters do not advance
    // (decoding.rs:484-520); it carries the counters of its POSITION (block_driver.pad_with_dummy_irs says why)
    rc = bp_ir_encode_dummy(o.block_number, chain->txn_number, chain->gas_used, chain->state_root, seed, log_n, width, w->ir);
  } else {
    uint64_t g0, g1;
    if ((rc = u256_low64(e.gas_used_before, &g0, "gas_used_before")) || (rc = u256_low64(e.gas_used_after, &g1, "gas_used_after"))) return rc;
    if (g1 < g0) return fail(BP_ERR_INVALID_INPUT, "generation inputs: gas_used_after < gas_used_before");
    rc = bp_ir_encode(o.block_number, chain->txn_number, chain->gas_used, chain->gas_used + (g1 - g0), chain->state_root, seed, log_n,
                      width, w->ir);
    if (rc == BP_OK) {
      uint64_t after[4];
      if ((rc = bp_state_root_after(chain->state_root, seed, chain->txn_number, after))) return rc;
      std::memcpy(chain->state_root, after, sizeof(after));
      chain->txn_number += 1;
      chain->gas_used += g1 - g0;
    }
  }
  if (rc) return rc;
  if ((f & BP_GI_KECCAK_AIR) && (rc = bp_ir_set_keccak_air(w->ir, 1))) return rc;
  if ((f & BP_GI_KECCAK_SPONGE_AIR) && (rc = bp_ir_set_keccak_sponge_air(w->ir, 1))) return rc;
  if ((f & BP_GI_MEMORY_AIR) && (rc = bp_ir_set_memory_air(w->ir, 1))) return rc;
  if ((f & BP_GI_BYTE_PACKING_AIR) && (rc = bp_ir_set_byte_packing_air(w->ir, 1))) return rc;
  if ((f & BP_GI_LOGIC_AIR) && (rc = bp_ir_set_logic_air(w->ir, 1))) return rc;
  return BP_OK;
}
int chain_start(const std::vector<Entry>& es, bp_gi_chain* chain) {
  if (es.empty()) return fail(BP_ERR_INVALID_INPUT, "generation inputs: no entries");
  mpt::Trie t;
  if (int rc = trie_of(es[0].tries[0], &t, "state")) return rc;
  const H256 h = t.hash();
  chain->txn_number = chain->gas_used = 0;
  for (int i = 0; i < 4; i++) chain->state_root[i] = le64(h.data() + 8 * i) % gl::P;
  return BP_OK;
}
int prove_entry(const bp_state* s, const Entry& e, const bp_gi_options& o, bp_gi_chain* chain, const volatile uint8_t* abort_flag,
                uint8_t** out, size_t* out_len) {
  EntryWork w;
  if (int rc = entry_work(e, o, chain, true, &w)) return rc;
  if (!w.any_witness) return bp_generate_txn_proof_u8(s, reinterpret_cast<const uint8_t*>(w.ir), sizeof(w.ir), abort_flag, out, out_len);
  bp_txn_witness tw;
  std::memset(&tw, 0, sizeof(tw));
  tw.keccak_inputs = w.perms.data(); tw.n_perms = w.perms.size() / 25; tw.has_keccak = 1;
  if (o.flags & BP_GI_KECCAK_SPONGE_AIR) { tw.sponge_rows = w.rows.data(); tw.n_sponge_rows = w.rows.size() / 44; tw.has_keccak_sponge = 1; }
  if (o.flags & BP_GI_MEMORY_AIR) { tw.memory_log = w.log.data(); tw.n_memory_ops = w.log.size() / 11; tw.has_memory = 1; }
  if (o.flags & BP_GI_BYTE_PACKING_AIR) { tw.byte_sequences = w.seqs.data(); tw.n_byte_sequences = w.seqs.size() / 6; tw.has_byte_packing = 1; }
  return bp_generate_txn_proof_witness(s, reinterpret_cast<const uint8_t*>(w.ir), sizeof(w.ir), &tw, abort_flag, out, out_len);
}

// ---------------------------------------------------------------- aggregation plan + scheduler
// Entry k of the plan is node n + k = (left, right) node ids; the last entry is the root.  Aggregation needs contiguous
// ranges (proof_types.rs:23-24) and nothing else.  0 = balanced: adjacent pairs level by level, an odd tail carried up;
// 1 = pairs_then_chain: adjacent leaves paired, the pair results folded left to right.
int make_plan(uint32_t n, uint32_t shape, std::vector<std::pair<uint32_t, uint32_t>>* plan) {
  if (n < 1) return fail(BP_ERR_INVALID_INPUT, "nothing to aggregate");
  if (shape > 1) return fail(BP_ERR_INVALID_INPUT, "unknown tree shape %u (0 balanced, 1 pairs_then_chain)", shape);
  plan->clear();
  if (shape == 0) {
    std::vector<uint32_t> level(n);
    for (uint32_t i = 0; i < n; i++) level[i] = i;
    while (level.size() > 1) {
      std::vector<uint32_t> nxt;
      for (size_t k = 0; k + 1 < level.size(); k += 2) {
        plan->push_back({level[k], level[k + 1]});
        nxt.push_back(n + (uint32_t)plan->size() - 1);
      }
      if (level.size() % 2) nxt.push_back(level.back());
      level.swap(nxt);
    }
    return BP_OK;
  }
  std::vector<uint32_t> heads;
  for (uint32_t k = 0; k + 1 < n; k += 2) {
    plan->push_back({k, k + 1});
    heads.push_back(n + (uint32_t)plan->size() - 1);
  }
  if (n % 2) heads.push_back(n - 1);
  uint32_t acc = heads[0];
  for (size_t i = 1; i < heads.size(); i++) {
    plan->push_back({acc, heads[i]});
    acc = n + (uint32_t)plan->size() - 1;
  }
  return BP_OK;
}

struct Buf {
  uint8_t* p = nullptr;
  size_t n = 0;
};
// All leaves of a contiguous slice and its tree.  Every aggregation starts the moment both of its children exist and goes
// AHEAD of the leaves still waiting for a thread (priority 0 before 1, each kind in index order), so the tree advances
// with the proving instead of piling up behind it.  The first failure stops the pool; its status and message are the call's.
// leaf_is_agg (nullable): the kinds of the leaves when they are proofs made elsewhere (bp_aggregate_proofs); else txn proofs
int run_tree(uint32_t n, const bp_shard_options* opt, bp_shard_leaf_fn leaf, bp_shard_agg_fn agg, void* ctx,
             const volatile uint8_t* abort_flag, uint8_t** root_out, size_t* root_len, uint8_t** leaf_out, size_t* leaf_len,
             const int* leaf_is_agg = nullptr) {
  if (!leaf || !agg || !root_out || !root_len) return fail(BP_ERR_INVALID_INPUT, "bp_run_shard: null argument");
  if ((leaf_out == nullptr) != (leaf_len == nullptr)) return fail(BP_ERR_INVALID_INPUT, "bp_run_shard: leaf_out and leaf_len go together");
  std::vector<std::pair<uint32_t, uint32_t>> plan;
  if (int rc = make_plan(n, opt ? opt->tree_shape : 0, &plan)) return rc;
  const uint32_t total = n + (uint32_t)plan.size(), root = total - 1;
  std::vector<uint32_t> parent_of(total, ~0u);
  for (uint32_t k = 0; k < plan.size(); k++) parent_of[plan[k].first] = parent_of[plan[k].second] = n + k;
  std::vector<Buf> res(total);
  std::vector<char> done(total, 0);
  std::mutex mu;
  std::condition_variable cv;
  using Item = std::pair<int, uint32_t>;
  std::priority_queue<Item, std::vector<Item>, std::greater<Item>> queue;
  for (uint32_t i = 0; i < n; i++) queue.push({1, i});
  bool stop = false;
  int first_rc = BP_OK;
  std::string first_msg;
  auto worker = [&] {
    for (;;) {
      uint32_t nid;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !queue.empty(); });
        if (stop) return;
        nid = queue.top().second;
        queue.pop();
      }
      Buf b;
      int rc;
      try {
        if (abort_flag && *abort_flag) rc = fail(BP_ERR_ABORTED, "aborted before node %u of the shard", nid);
        else if (nid < n) rc = leaf(ctx, nid, &b.p, &b.n);
        else {
          const uint32_t l = plan[nid - n].first, r = plan[nid - n].second;
          const int la = l >= n || (leaf_is_agg && leaf_is_agg[l]), ra = r >= n || (leaf_is_agg && leaf_is_agg[r]);
          rc = agg(ctx, res[l].p, res[l].n, la, res[r].p, res[r].n, ra, &b.p, &b.n);
        }
        if (rc == BP_OK && !b.p) rc = fail(BP_ERR_DEVICE, "node %u of the shard returned no proof", nid);
      } catch (...) {
        rc = fail(BP_ERR_DEVICE, "node %u of the shard: exception in a callback", nid);
      }
      std::lock_guard<std::mutex> lk(mu);
      if (rc) {
        if (!first_rc) { first_rc = rc; first_msg = bp_last_error(); }
        std::free(b.p);
        stop = true;
        cv.notify_all();
        return;
      }
      res[nid] = b;
      done[nid] = 1;
      if (nid >= n) {  // the children have been consumed: leaves stay when the caller wants them
        const uint32_t ch[2] = {plan[nid - n].first, plan[nid - n].second};
        for (uint32_t c : ch)
          if (c >= n || !leaf_out) { std::free(res[c].p); res[c] = Buf(); }
      }
      const uint32_t par = parent_of[nid];
      if (par != ~0u && done[plan[par - n].first] && done[plan[par - n].second]) queue.push({0, par});
      if (nid == root) stop = true;
      cv.notify_all();
    }
  };
  uint32_t n_threads = opt && opt->n_threads ? opt->n_threads : 1;
  n_threads = std::min<uint32_t>(std::min<uint32_t>(n_threads, n), 256);
  std::vector<std::thread> pool;
  struct Joiner {
    std::vector<std::thread>& p;
    ~Joiner() { for (auto& t : p) if (t.joinable()) t.join(); }
  } joiner{pool};
  for (uint32_t i = 1; i < n_threads; i++) {
    try {
      pool.emplace_back(worker);
    } catch (const std::system_error&) {  // no more threads to be had: the ones there are do the work
      break;
    }
  }
  worker();  // the calling thread is one of the pool
  for (auto& t : pool) t.join();
  pool.clear();
  if (first_rc) {
    for (auto& b : res) std::free(b.p);
    return fail(first_rc, "%s", first_msg.c_str());
  }
  *root_out = res[root].p;
  *root_len = res[root].n;
  if (leaf_out) {
    for (uint32_t i = 0; i < n; i++) {
      if (n == 1) {  // the root IS the leaf: the caller gets its own copy to free
        leaf_out[0] = static_cast<uint8_t*>(std::malloc(res[0].n));
        if (!leaf_out[0]) { std::free(res[0].p); *root_out = nullptr; return fail(BP_ERR_DEVICE, "host allocation failed"); }
        std::memcpy(leaf_out[0], res[0].p, res[0].n);
        leaf_len[0] = res[0].n;
      } else {
        leaf_out[i] = res[i].p;
        leaf_len[i] = res[i].n;
      }
    }
  }
  return BP_OK;
}

struct IrShard {
  const bp_state* s;
  const uint8_t* irs;
  size_t stride;
  const volatile uint8_t* abort_flag;
};
int ir_leaf(void* ctx, uint32_t i, uint8_t** out, size_t* out_len) {
  const IrShard* c = static_cast<const IrShard*>(ctx);
  return bp_generate_txn_proof_u8(c->s, c->irs + (size_t)i * c->stride, BP_IR_WORDS * 8, c->abort_flag, out, out_len);
}
int state_agg(void* ctx, const uint8_t* l, size_t ln, int l_agg, const uint8_t* r, size_t rn, int r_agg, uint8_t** out, size_t* out_len) {
  return bp_generate_agg_proof(*static_cast<const bp_state* const*>(ctx), l, ln, l_agg, r, rn, r_agg, out, out_len);
}
struct GiShard {
  const bp_state* s;  // first member: state_agg reads it through the same pointer
  const std::vector<Entry>* es;
  const bp_gi_options* o;
  uint32_t first;
  const std::vector<bp_gi_chain>* chain_before;
  const volatile uint8_t* abort_flag;
};
int gi_leaf(void* ctx, uint32_t i, uint8_t** out, size_t* out_len) {
  const GiShard* c = static_cast<const GiShard*>(ctx);
  bp_gi_chain ch = (*c->chain_before)[c->first + i];
  return prove_entry(c->s, (*c->es)[c->first + i], *c->o, &ch, c->abort_flag, out, out_len);
}
uint32_t default_threads(const bp_state* s, const bp_shard_options* opt) {
  if (opt && opt->n_threads) return opt->n_threads;
  bp_config cfg;
  return bp_state_config(s, &cfg) == BP_OK ? cfg.n_workers : 1;
}

}  // namespace

extern "C" {

int bp_gi_count(const uint8_t* geni, size_t len, uint32_t* n_entries) try {
  std::vector<Entry> es;
  if (int rc = parse_geni(geni, len, &es)) return rc;
  if (n_entries) *n_entries = (uint32_t)es.size();
  return BP_OK;
}
BPG_ABI_CATCH("bp_gi_count")

int bp_gi_chain_start(const uint8_t* geni, size_t len, bp_gi_chain* chain) try {
  if (!chain) return fail(BP_ERR_INVALID_INPUT, "bp_gi_chain_start: null argument");
  std::vector<Entry> es;
  if (int rc = parse_geni(geni, len, &es)) return rc;
  return chain_start(es, chain);
}
BPG_ABI_CATCH("bp_gi_chain_start")

int bp_gi_entry_ir(const uint8_t* geni, size_t len, uint32_t entry, const bp_gi_options* opt, bp_gi_chain* chain,
                   uint64_t ir_out[BP_IR_WORDS]) try {
  if (!opt || !chain || !ir_out) return fail(BP_ERR_INVALID_INPUT, "bp_gi_entry_ir: null argument");
  std::vector<Entry> es;
  if (int rc = parse_geni(geni, len, &es)) return rc;
  if (entry >= es.size()) return fail(BP_ERR_INVALID_INPUT, "entry %u of %zu", entry, es.size());
  EntryWork w;
  if (int rc = entry_work(es[entry], *opt, chain, false, &w)) return rc;
  std::memcpy(ir_out, w.ir, sizeof(w.ir));
  return BP_OK;
}
BPG_ABI_CATCH("bp_gi_entry_ir")

int bp_generate_txn_proof_gi(const bp_state* s, const uint8_t* geni, size_t len, uint32_t entry, const bp_gi_options* opt,
                             bp_gi_chain* chain, const volatile uint8_t* abort_flag, uint8_t** out, size_t* out_len) try {
  if (!s || !opt || !chain || !out || !out_len) return fail(BP_ERR_INVALID_INPUT, "bp_generate_txn_proof_gi: null argument");
  std::vector<Entry> es;
  if (int rc = parse_geni(geni, len, &es)) return rc;
  if (entry >= es.size()) return fail(BP_ERR_INVALID_INPUT, "entry %u of %zu", entry, es.size());
  bp_gi_chain after = *chain;
  if (int rc = prove_entry(s, es[entry], *opt, &after, abort_flag, out, out_len)) return rc;
  *chain = after;
  return BP_OK;
}
BPG_ABI_CATCH("bp_generate_txn_proof_gi")

uint32_t bp_aggregation_plan(uint32_t n, uint32_t shape, uint32_t* pairs) try {
  std::vector<std::pair<uint32_t, uint32_t>> plan;
  if (make_plan(n, shape, &plan)) return ~0u;
  if (pairs)
    for (size_t k = 0; k < plan.size(); k++) { pairs[2 * k] = plan[k].first; pairs[2 * k + 1] = plan[k].second; }
  return (uint32_t)plan.size();
} catch (...) { return ~0u; }

int bp_run_shard(uint32_t n, const bp_shard_options* opt, bp_shard_leaf_fn leaf, bp_shard_agg_fn agg, void* ctx,
                 const volatile uint8_t* abort_flag, uint8_t** root_out, size_t* root_len, uint8_t** leaf_out, size_t* leaf_len) try {
  return run_tree(n, opt, leaf, agg, ctx, abort_flag, root_out, root_len, leaf_out, leaf_len);
}
BPG_ABI_CATCH("bp_run_shard")

// The top of a block's tree: n proofs made elsewhere (the sub-block proofs gathered from the other ranks; txn or
// aggregation proofs, contiguous in this order) folded into one along the same plan, aggregations that do not depend on
// each other side by side.
int bp_aggregate_proofs(const bp_state* s, const uint8_t* const* proofs, const size_t* lens, uint32_t n, const bp_shard_options* opt,
                        uint8_t** out, size_t* out_len) try {
  if (!s || !proofs || !lens || !out || !out_len) return fail(BP_ERR_INVALID_INPUT, "bp_aggregate_proofs: null argument");
  std::vector<int> kinds(n);
  for (uint32_t i = 0; i < n; i++) {
    int kind = 0;
    if (int rc = bp_proof_public_values(proofs[i], lens[i], nullptr, &kind)) return rc;
    if (kind > 1) return fail(BP_ERR_INVALID_INPUT, "proof %u is a block proof: only txn and aggregation proofs aggregate", i);
    kinds[i] = kind;
  }
  struct Ctx { const bp_state* s; const uint8_t* const* proofs; const size_t* lens; } c{s, proofs, lens};
  auto leaf = [](void* ctx, uint32_t i, uint8_t** o, size_t* ol) {
    const Ctx* x = static_cast<const Ctx*>(ctx);
    *o = static_cast<uint8_t*>(std::malloc(x->lens[i] ? x->lens[i] : 1));
    if (!*o) return fail(BP_ERR_DEVICE, "host allocation failed");
    std::memcpy(*o, x->proofs[i], x->lens[i]);
    *ol = x->lens[i];
    return (int)BP_OK;
  };
  bp_shard_options o{default_threads(s, opt), opt ? opt->tree_shape : 0};
  return run_tree(n, &o, leaf, state_agg, &c, nullptr, out, out_len, nullptr, nullptr, kinds.data());
}
BPG_ABI_CATCH("bp_aggregate_proofs")

int bp_prove_shard(const bp_state* s, const uint8_t* irs, size_t ir_stride, uint32_t n, const bp_shard_options* opt,
                   const volatile uint8_t* abort_flag, uint8_t** root_out, size_t* root_len, uint8_t** txn_out, size_t* txn_len) try {
  if (!s || !irs) return fail(BP_ERR_INVALID_INPUT, "bp_prove_shard: null argument");
  if (ir_stride < BP_IR_WORDS * 8) return fail(BP_ERR_INVALID_INPUT, "bp_prove_shard: ir_stride below %d bytes", BP_IR_WORDS * 8);
  struct Ctx { const bp_state* s; IrShard sh; } c{s, {s, irs, ir_stride, abort_flag}};
  bp_shard_options o{default_threads(s, opt), opt ? opt->tree_shape : 0};
  auto leaf = [](void* ctx, uint32_t i, uint8_t** out, size_t* out_len) { return ir_leaf(&static_cast<Ctx*>(ctx)->sh, i, out, out_len); };
  return run_tree(n, &o, leaf, state_agg, &c, abort_flag, root_out, root_len, txn_out, txn_len);
}
BPG_ABI_CATCH("bp_prove_shard")

int bp_prove_shard_gi(const bp_state* s, const uint8_t* geni, size_t len, uint32_t first, uint32_t n, const bp_gi_options* gi,
                      const bp_shard_options* opt, const volatile uint8_t* abort_flag, uint8_t** root_out, size_t* root_len,
                      uint8_t** txn_out, size_t* txn_len) try {
  if (!s || !gi) return fail(BP_ERR_INVALID_INPUT, "bp_prove_shard_gi: null argument");
  std::vector<Entry> es;
  if (int rc = parse_geni(geni, len, &es)) return rc;
  if (first > es.size() || n > es.size() - first) return fail(BP_ERR_INVALID_INPUT, "entries %u..%u of %zu", first, first + n, es.size());
  // the chain before every entry of the slice: counters and state root threaded through the entries before it (cheap
  // host hashing, sequential by nature: decoding.rs:106-154)
  std::vector<bp_gi_chain> before(es.size());
  bp_gi_chain ch;
  if (int rc = chain_start(es, &ch)) return rc;
  for (uint32_t k = 0; k < first + n; k++) {
    before[k] = ch;
    EntryWork w;
    if (int rc = entry_work(es[k], *gi, &ch, false, &w)) return rc;
  }
  GiShard c{s, &es, gi, first, &before, abort_flag};
  bp_shard_options o{default_threads(s, opt), opt ? opt->tree_shape : 0};
  return run_tree(n, &o, gi_leaf, state_agg, &c, abort_flag, root_out, root_len, txn_out, txn_len);
}
BPG_ABI_CATCH("bp_prove_shard_gi")

}  // extern "C"
