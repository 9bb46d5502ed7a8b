// proofgen.cpp -- L1 of the C ABI: prover/verifier state and the txn / agg / block entry points,
// mirroring plonky_block_proof_gen/src/{prover_state.rs, proof_gen.rs, verifier_state.rs,
// proof_types.rs}.  What each call computes is the synthetic workload of SURVEY.md section 8(d)
// (DESIGN.md section 5); the control flow, ownership, threading and error behaviour follow the
// reference (see include/bpg.h).
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <deque>
#include <set>
#include "mpt.hpp"
#include "prover.hpp"

using namespace bpg;
extern "C" int bp_use_blocking_sync(int device);
extern "C" int bp_host_wait_mode(int device);

namespace {

constexpr uint64_t IR_MAGIC = 0x52494E5854475042ULL;     // "BPGTXNIR"
constexpr uint64_t PROOF_BOX_MAGIC = 0x464F4F5250475042ULL;  // "BPGPROOF"
constexpr uint64_t TABLES_MAGIC = 0x534C424154475042ULL;     // "BPGTABLS"
constexpr uint32_t CIRCUIT_ROOT = 7, CIRCUIT_AGG = 8, CIRCUIT_BLOCK = 9;
constexpr uint32_t AGG_PATH_PI0 = 10, BLOCK_PATH_PI0 = 9;  // where the children's (leaf digest, cap entry) words sit in the list
constexpr uint32_t CHAIN_PATH_PI0 = 6, ROOT_PATH_PI0 = 4 * BP_NUM_TABLES;  // ... of a chain circuit's one child, of the root circuit's seven
constexpr uint32_t SHRINK_SEED_DEGREE = 255;  // circuit_seed(table, 255): the table's shrink circuit (levels >= 1 of its chain)
constexpr size_t BOX_HDR = 4;
const char* TABLE_NAMES[BP_NUM_TABLES] = {"arithmetic", "byte_packing", "cpu", "keccak", "keccak_sponge", "logic", "memory"};

uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
uint64_t circuit_seed(uint32_t table_or_kind, uint32_t degree) {
  return splitmix64(0xC12C0175EEDULL + ((uint64_t)table_or_kind << 8) + degree);
}

struct Circuit {  // one preprocessed recursion circuit: constants commitment + its digest
  uint64_t* d_const_values = nullptr;
  Committed consts;
  uint64_t digest[4];
  air::plonk::Layout lay{};  // AIR 8: the list the circuit hashes and the Merkle paths it walks
};
// One Merkle path a recursion circuit walks in its Poseidon rows: the leaf digest and the cap entry are words of the
// proof's public-input list (Layout::path_pi0), the position and the siblings are witness.
struct PathWitness {
  uint64_t index = 0;
  std::vector<uint64_t> siblings;  // 4 words per level, leaf upward
  std::vector<uint64_t> leaf_row;  // the opened row the leaf digest is the hash of (circuits that hash it: Layout::leaf_len)
};
struct LightCircuit {
  std::vector<uint64_t> cap;
  uint64_t digest[4];
};

}  // namespace

struct bp_state {
  bp_config cfg;
  StarkCfg rec_cfg;
  Worker builder;                       // owns the persistent (preprocessed) device memory
  std::vector<Circuit> table_circuits;  // [table][degree - lo] flattened: level 0 of a table's chain (its child is the table's STARK proof)
  Circuit shrink_circuits[BP_NUM_TABLES];  // levels >= 1 of a table's chain (their child is a recursion-shaped proof)
  uint32_t table_offset[BP_NUM_TABLES];
  Circuit special[3];                   // root, agg, block
  // worker pool: `&ProverState` is shared by many threads in the reference (proof_gen.rs:40)
  mutable std::mutex mu;
  mutable std::condition_variable cv;
  mutable std::vector<std::unique_ptr<Worker>> workers;
  mutable std::vector<Worker*> idle;
  // Keccak-256 of every proof container this state has produced (bounded): aggregation verifies its children on the
  // host (verify_child), which costs 10 ms of CPU per recursion-shaped proof against 5 ms of GPU to make one -- a child
  // that this very state has just produced, byte for byte, is recognised instead of being verified again
  mutable std::mutex seen_mu;
  mutable std::set<mpt::H256> seen;
  mutable std::deque<mpt::H256> seen_order;
  std::string warnings;  // bp_state_warnings: what bp_state_build found about the environment it runs in
};
struct bp_verifier_state {
  StarkCfg rec_cfg;
  LightCircuit special[3];
};

namespace {

struct WorkerLease {
  const bp_state* s;
  Worker* w;
  size_t mark;
  explicit WorkerLease(const bp_state* st) : s(st) {
    std::unique_lock<std::mutex> lk(s->mu);
    s->cv.wait(lk, [&] { return !s->idle.empty(); });
    w = s->idle.back();
    s->idle.pop_back();
    mark = w->arena.mark();
    prover_active(+1);
  }
  ~WorkerLease() {
    prover_active(-1);
    (void)hipStreamSynchronize(w->stream);
    w->arena.release(mark);
    w->abort_flag = nullptr;
    w->abort_flag_u8 = nullptr;
    std::lock_guard<std::mutex> lk(s->mu);
    s->idle.push_back(w);
    s->cv.notify_one();
  }
};

// An idle worker borrowed as a LANE -- its stream and its cap mailbox, nothing else -- by a prover that finds itself
// ALONE on the device (a lone transaction, the last one of a shard): work that does not depend on each other (the seven
// trace commitments of a transaction) then overlaps instead of running one medium launch after the other.  Never waits
// for a worker, is not counted as a prover.  (Streams of their own for this were measured and dropped: three more streams
// in the process cost the loaded 256-txn block 2.6 % whether they were used or not -- 24 streams instead of 21, the same
// step the stream sweep shows between 20 and 24 prover streams.)
struct SideLane {
  const bp_state* s;
  Worker* w;
  static std::unique_ptr<SideLane> try_acquire(const bp_state* st) {
    std::lock_guard<std::mutex> lk(st->mu);
    if (st->idle.empty()) return nullptr;
    std::unique_ptr<SideLane> l(new SideLane{st, st->idle.back()});
    st->idle.pop_back();
    return l;
  }
  ~SideLane() {
    (void)hipStreamSynchronize(w->stream);  // nothing of ours is left on the lane when its owner gets it back
    std::lock_guard<std::mutex> lk(s->mu);
    s->idle.push_back(w);
    s->cv.notify_one();
  }
};

StarkCfg rec_cfg_of(const bp_config& c) {
  return StarkCfg{c.rec_log_n, c.rec_n_cols, c.rec_n_const, 3, c.rec_rate_bits, c.stark_cap_height,
                  c.rec_num_queries, c.rec_pow_bits, c.arity_bits, c.final_poly_bits, c.rec_air_id};
}
StarkCfg table_cfg_of(const bp_config& c, uint32_t log_n, uint32_t width) {
  return StarkCfg{log_n, width, 0, 1, c.stark_rate_bits, c.stark_cap_height, c.stark_num_queries,
                  c.stark_pow_bits, c.arity_bits, c.final_poly_bits};
}

// lay (AIR 8): the public-input list the circuit's hash rows absorb -- 6 words for a table's chain circuits (digest,
// table, depth), 7 x 4 + 13 for the root circuit, 10 + 2 x 8 + 13 / 9 + 8 + 13 for the aggregation / block circuit -- and
// the Merkle paths it walks: one per child proof of the aggregation circuit, the aggregation child's of the block circuit
int build_circuit(Worker& w, const StarkCfg& rc, uint64_t seed, const air::plonk::Layout& lay, Circuit* out) {
  const uint64_t N = (uint64_t)1 << rc.log_n;
  out->lay = lay;
  out->d_const_values = w.arena.alloc_words((size_t)rc.n_const * N);
  if (!out->d_const_values) return fail(BP_ERR_DEVICE, "state arena exhausted");
  int rc2 = rc.air_id == air::PLONK ? launch_plonk_constants(out->d_const_values, rc.log_n, seed, lay, w.stream)
                                    : launch_synth_constants(out->d_const_values, rc.log_n, rc.n_const, seed, w.stream);
  if (rc2) return rc2;
  if ((rc2 = commit(w, out->d_const_values, rc.n_const, rc.log_n, rc.rate_bits, rc.cap_height, false, &out->consts)))
    return rc2;
  hash_no_pad_host(out->consts.cap.data(), out->consts.cap.size(), out->digest);
  return BP_OK;
}

struct Box {  // parsed proof container
  uint64_t kind, n_pi, circuit;
  const uint64_t *pi, *pv, *stark;
  size_t stark_words;
};
int parse_box(const uint8_t* bytes, size_t len, const StarkCfg& rc, Box* b) {
  if (!bytes || len % 8 || len < (BOX_HDR + BP_PV_WORDS) * 8) return fail(BP_ERR_INVALID_INPUT, "proof: truncated");
  const uint64_t* wds = reinterpret_cast<const uint64_t*>(bytes);
  if (wds[0] != PROOF_BOX_MAGIC) return fail(BP_ERR_INVALID_INPUT, "proof: bad magic");
  b->kind = wds[1]; b->n_pi = wds[2]; b->circuit = wds[3];
  if (b->kind > 2 || b->n_pi < BP_PV_WORDS || b->n_pi > air::plonk::MAX_PI) return fail(BP_ERR_INVALID_INPUT, "proof: bad header");
  const size_t sw = proof_layout(rc).total;
  if (len / 8 != BOX_HDR + b->n_pi + sw) return fail(BP_ERR_INVALID_INPUT, "proof: wrong length for this circuit");
  b->pi = wds + BOX_HDR;
  b->pv = b->pi + b->n_pi - BP_PV_WORDS;
  b->stark = b->pi + b->n_pi;
  b->stark_words = sw;
  return BP_OK;
}
int emit_box(uint64_t kind, uint64_t circuit, const std::vector<uint64_t>& pi, const std::vector<uint64_t>& stark,
             uint8_t** out, size_t* out_len) {
  const size_t words = BOX_HDR + pi.size() + stark.size();
  uint64_t* o = static_cast<uint64_t*>(std::malloc(words * 8));
  if (!o) return fail(BP_ERR_DEVICE, "host allocation failed");
  o[0] = PROOF_BOX_MAGIC; o[1] = kind; o[2] = pi.size(); o[3] = circuit;
  std::memcpy(o + BOX_HDR, pi.data(), pi.size() * 8);
  std::memcpy(o + BOX_HDR + pi.size(), stark.data(), stark.size() * 8);
  *out = reinterpret_cast<uint8_t*>(o);
  *out_len = words * 8;
  return BP_OK;
}

// Recursion-shaped proofs; transcript of each = circuit digest, hash of the public inputs, trace cap.  `n` proofs of
// the state's one recursion shape are proved in lock-step, up to g_rec_batch at a time (stark_prove_batch: every
// launch and every host wait is shared; circuits, public inputs and transcripts are each proof's own).
std::atomic<uint32_t> g_rec_batch{MAX_BATCH};  // bp_tune_rec_batch: 1 = one proof at a time
std::atomic<int> g_witness_threads{7};           // bp_tune_witness_threads: host threads a lone prover makes its Poseidon-row witness on
std::atomic<int> g_side_lanes{1};               // bp_tune_side_lanes: 0 = no side lanes, n = while at most n provers are at work
// paths (nullable): per proof the witness of the Merkle paths its circuit walks (Circuit::lay.n_paths of them)
// first_leaf (nullable, 4 words per proof): the digest of the trace leaf each proof's first query opens
int rec_prove_batch(Worker& w, const StarkCfg& rc, uint32_t n, const Circuit* const* circ, const std::vector<uint64_t>* pi,
                    std::vector<uint64_t>* proofs, const std::vector<PathWitness>* paths = nullptr, uint64_t* first_leaf = nullptr) {
  const uint32_t cap = std::min<uint32_t>(std::min<uint32_t>(MAX_BATCH, std::max<uint32_t>(1, g_rec_batch.load(std::memory_order_relaxed))),
                                          std::max<uint32_t>(1, MAX_BATCH_QUERIES / std::max<uint32_t>(1, rc.num_queries)));
  const uint64_t N = (uint64_t)1 << rc.log_n;
  for (uint32_t first = 0; first < n; first += cap) {
    const uint32_t B = std::min(cap, n - first);
    const size_t mark = w.arena.mark();
    uint64_t* d_trace = w.arena.alloc_words((size_t)B * rc.n_cols * N);
    if (!d_trace) return fail(BP_ERR_DEVICE, "device arena exhausted (%zu MiB)", w.arena.capacity() >> 20);
    Challenger ch[MAX_BATCH];
    SynthTraceArgs sa[MAX_BATCH];
    PlonkTraceArgs pa[MAX_BATCH];
    Ctl ctl[MAX_BATCH];
    const uint64_t* d_tv[MAX_BATCH];
    const Committed* consts[MAX_BATCH];
    // The witness of every proof's Poseidon rows is made on the host (prover.hpp): per proof the sponge over its list, per
    // path the Merkle rows and -- where the circuit hashes its leaves -- the sponge over the opened row.  The pieces are
    // independent; a prover that is alone on the device (a lone transaction: this is its critical path) makes them on up
    // to seven threads.
    uint64_t pi_hashes[MAX_BATCH][4];
    struct Job {
      uint32_t b, kind, p;  // kind 0: the list's rows, 1: path p's Merkle rows, 2: path p's leaf rows
    };
    std::vector<Job> jobs;
    for (uint32_t b = 0; b < B; b++) {
      const Circuit& c = *circ[first + b];
      const size_t n_pi = pi[first + b].size();
      if (rc.air_id != air::PLONK) {
        hash_no_pad_host(pi[first + b].data(), n_pi, pi_hashes[b]);
        continue;
      }
      if (n_pi < 1 || n_pi > air::plonk::MAX_PI) return fail(BP_ERR_INVALID_INPUT, "a recursion circuit hashes 1..%u public inputs: got %zu", air::plonk::MAX_PI, n_pi);
      if (n_pi != c.lay.pi_len) return fail(BP_ERR_INVALID_INPUT, "this circuit hashes a list of %u public inputs: got %zu", c.lay.pi_len, n_pi);
      jobs.push_back(Job{b, 0, 0});
      if (c.lay.n_paths) {
        if (!paths || paths[first + b].size() != c.lay.n_paths) return fail(BP_ERR_INVALID_INPUT, "this circuit walks %u Merkle paths: their witness is missing", c.lay.n_paths);
        for (uint32_t p = 0; p < c.lay.n_paths; p++) {
          const PathWitness& pw = paths[first + b][p];
          if (pw.siblings.size() != 4 * (size_t)c.lay.depth) return fail(BP_ERR_INVALID_INPUT, "Merkle path %u: %zu sibling words for %u levels", p, pw.siblings.size(), c.lay.depth);
          if (c.lay.leaf_len && pw.leaf_row.size() != c.lay.leaf_len) return fail(BP_ERR_INVALID_INPUT, "Merkle path %u: the circuit hashes a leaf of %u words, %zu given", p, c.lay.leaf_len, pw.leaf_row.size());
          jobs.push_back(Job{b, 1, p});
          if (c.lay.leaf_len) jobs.push_back(Job{b, 2, p});
        }
      }
    }
    auto run_job = [&](const Job& j) {
      const Circuit& c = *circ[first + j.b];
      uint64_t* hr = w.hash_rows + (size_t)j.b * HASH_ROWS_WORDS;
      std::vector<uint64_t> rows;
      if (j.kind == 0) {  // the hash of the public-input list, with the witness of the circuit's own computation of it
        poseidon_hash_rows(pi[first + j.b].data(), pi[first + j.b].size(), &rows, pi_hashes[j.b]);
        std::memcpy(hr, rows.data(), rows.size() * 8);
      } else if (j.kind == 1) {  // a child's Merkle path, walked the way the circuit's rows walk it
        const PathWitness& pw = paths[first + j.b][j.p];
        uint64_t root[4];  // (not compared with the list's cap entry here: the copy constraints do that, and the verifier)
        poseidon_merkle_rows(&pi[first + j.b][c.lay.path_pi0 + 8 * j.p], pw.index, pw.siblings.data(), c.lay.depth,
                             hr + (size_t)(air::plonk::HASH_ROWS_MAX + j.p * c.lay.depth) * air::plonk::H_WIRES, root);
      } else {  // the sponge over the opened row that the leaf digest is the hash of
        const PathWitness& pw = paths[first + j.b][j.p];
        uint64_t leaf_digest[4];
        poseidon_hash_rows(pw.leaf_row.data(), c.lay.leaf_len, &rows, leaf_digest);
        std::memcpy(hr + (size_t)(air::plonk::HASH_ROWS_MAX + air::plonk::merkle_rows(c.lay) + j.p * air::plonk::hash_rows(c.lay.leaf_len)) * air::plonk::H_WIRES,
                    rows.data(), rows.size() * 8);
      }
    };
    {
      const uint32_t max_threads = (uint32_t)std::max(1, g_witness_threads.load(std::memory_order_relaxed));
      const uint32_t n_threads = (provers_active() <= 1 && jobs.size() >= 4) ? (uint32_t)std::min<size_t>(max_threads, jobs.size()) : 1;
      std::atomic<size_t> next{0};
      std::atomic<bool> oom{false};
      auto drain = [&] {
        try {
          for (size_t k; (k = next.fetch_add(1)) < jobs.size();) run_job(jobs[k]);
        } catch (...) {
          oom.store(true);
        }
      };
      std::vector<std::thread> pool;
      struct Join {
        std::vector<std::thread>& p;
        ~Join() { for (auto& t : p) if (t.joinable()) t.join(); }
      } join{pool};
      try {
        for (uint32_t t = 1; t < n_threads; t++) pool.emplace_back(drain);
      } catch (const std::system_error&) {  // no thread to be had: this one does the rest
      }
      drain();
      for (auto& t : pool) t.join();
      pool.clear();
      if (oom.load()) return fail(BP_ERR_DEVICE, "out of memory while making the witness of the Poseidon rows");
    }
    for (uint32_t b = 0; b < B; b++) {
      const Circuit& c = *circ[first + b];
      const uint64_t* pi_hash = pi_hashes[b];
      const size_t n_pi = pi[first + b].size();
      ch[b].observe(c.digest, 4);
      ch[b].observe(pi_hash, 4);
      d_tv[b] = d_trace + (size_t)b * rc.n_cols * N;
      sa[b] = SynthTraceArgs{d_trace + (size_t)b * rc.n_cols * N, c.d_const_values, pi_hash[0]};
      pa[b] = PlonkTraceArgs{d_trace + (size_t)b * rc.n_cols * N, c.d_const_values, pi_hash[0], {pi_hash[0], pi_hash[1], pi_hash[2], pi_hash[3]},
                             w.hash_rows_dev + (size_t)b * HASH_ROWS_WORDS, (uint32_t)((n_pi + 7) / 8),
                             air::plonk::merkle_rows(c.lay) + air::plonk::leaf_rows(c.lay), air::plonk::arith_row0(c.lay)};
      if (rc.air_id == air::PLONK) std::memcpy(ctl[b].pub, pi_hash, 32);  // bound to the circuit's first row
      consts[b] = &c.consts;
    }
    int r = rc.air_id == air::PLONK ? launch_plonk_trace(pa, B, rc.log_n, w.stream)
                                    : launch_synth_trace(sa, B, rc.log_n, rc.n_cols, rc.n_const, rc.deg_pow, w.stream);
    if (r) return r;
    Committed trace[MAX_BATCH];
    if ((r = commit_batch(w, d_trace, rc.n_cols, B, rc.log_n, rc.rate_bits, rc.cap_height, false, trace))) return r;
    for (uint32_t b = 0; b < B; b++) {
      ch[b].observe(trace[b].cap.data(), trace[b].cap.size());
      for (int i = 0; i < 4; i++) ctl[b].v[i] = ch[b].challenge();
    }
    r = stark_prove_batch(w, rc, B, consts, trace, d_tv, ctl, ch, proofs + first, first_leaf ? first_leaf + 4 * first : nullptr);
    w.arena.release(mark);
    if (r) return r;
  }
  return BP_OK;
}
int rec_prove(Worker& w, const StarkCfg& rc, const Circuit& circ, const std::vector<uint64_t>& pi,
              std::vector<uint64_t>& proof, const std::vector<PathWitness>* paths = nullptr) {
  const Circuit* c = &circ;
  return rec_prove_batch(w, rc, 1, &c, &pi, &proof, paths);
}
// What the aggregation / block circuit walks for a child: the Merkle path of the child's FIRST query into its trace
// oracle.  leaf = the digest of the opened trace row, cap_entry = the entry of the child's trace cap the path ends in
// (both become words of the parent's public-input list), pw = position and siblings.  `child` has been parsed
// (parse_box: its length is the layout's).
// known_leaf (nullable): the leaf digest as the prover's device gave it (stark_prove's first_trace_leaf) -- a child made
// elsewhere has its opened row hashed here.
void first_query_trace_path(const StarkCfg& c, const uint64_t* child, uint64_t leaf[4], uint64_t cap_entry[4], PathWitness* pw,
                            const uint64_t* known_leaf = nullptr) {
  const ProofLayout L = proof_layout(c);
  const uint64_t* w = child + L.queries;
  const uint64_t x = *w++;
  if (c.n_const) w += c.n_const + (size_t)L.depth0 * 4;
  pw->leaf_row.assign(w, w + c.n_cols);
  if (known_leaf) {
    std::memcpy(leaf, known_leaf, 32);
  } else if (c.n_cols <= 4) {  // Hasher::hash_or_noop
    std::memset(leaf, 0, 32);
    std::memcpy(leaf, w, c.n_cols * 8);
  } else {
    hash_no_pad_host(w, c.n_cols, leaf);
  }
  w += c.n_cols;
  pw->index = x & (((uint64_t)1 << L.depth0) - 1);
  pw->siblings.assign(w, w + 4 * (size_t)L.depth0);
  const uint64_t top = (x >> L.depth0) & (((uint64_t)1 << c.cap_height) - 1);
  std::memcpy(cap_entry, child + L.trace_cap + 4 * top, 32);
}
// the length of the public-input list each kind of container carries (what its circuit hashes in its thirteen rows at most)
uint32_t box_pi_len(uint64_t kind) {
  return kind == 0 ? ROOT_PATH_PI0 + 8 * BP_NUM_TABLES + BP_PV_WORDS : kind == 1 ? AGG_PATH_PI0 + 16 + BP_PV_WORDS : BLOCK_PATH_PI0 + 8 + BP_PV_WORDS;
}
int rec_verify(const StarkCfg& rc, const LightCircuit& circ, const Box& b) {
  if (b.n_pi != box_pi_len(b.kind))
    return fail(BP_ERR_VERIFY, "a proof of kind %llu carries %u public inputs, this one %llu", (unsigned long long)b.kind, box_pi_len(b.kind),
                (unsigned long long)b.n_pi);
  uint64_t pi_hash[4];
  hash_no_pad_host(b.pi, b.n_pi, pi_hash);
  Challenger ch;
  ch.observe(circ.digest, 4);
  ch.observe(pi_hash, 4);
  const ProofLayout L = proof_layout(rc);
  ch.observe(b.stark + L.trace_cap, L.cap_words);
  Ctl ctl;
  for (int i = 0; i < 4; i++) ctl.v[i] = ch.challenge();
  if (rc.air_id == air::PLONK) std::memcpy(ctl.pub, pi_hash, sizeof(pi_hash));
  return stark_verify(rc, circ.cap.data(), ctl, ch, b.stark, b.stark_words);
}

// What the real aggregation / block circuits do in-circuit (prove_aggregation / prove_block verify their
// children, proof_gen.rs:66-75, 97-103) is done on the host here: a child container is accepted only if
// it was made by the circuit its kind names, its public inputs are canonical and its proof verifies
// against this state's preprocessed circuit.
constexpr size_t SEEN_MAX = 8192;
void remember_proof(const bp_state* s, const uint8_t* bytes, size_t len) {
  const mpt::H256 h = mpt::keccak256(bytes, len);
  std::lock_guard<std::mutex> lk(s->seen_mu);
  if (!s->seen.insert(h).second) return;
  s->seen_order.push_back(h);
  if (s->seen_order.size() > SEEN_MAX) {
    s->seen.erase(s->seen_order.front());
    s->seen_order.pop_front();
  }
}
bool produced_here(const bp_state* s, const uint8_t* bytes, size_t len) {
  const mpt::H256 h = mpt::keccak256(bytes, len);
  std::lock_guard<std::mutex> lk(s->seen_mu);
  return s->seen.count(h) != 0;
}

int verify_foreign(const bp_state* s, const Box& b, const char* what) {
  if (b.circuit != CIRCUIT_ROOT + b.kind)
    return fail(BP_ERR_VERIFY, "%s was made by circuit %llu, expected %u", what, (unsigned long long)b.circuit,
                CIRCUIT_ROOT + (uint32_t)b.kind);
  for (size_t i = 0; i < b.n_pi; i++)
    if (b.pi[i] >= gl::P) return fail(BP_ERR_VERIFY, "%s: non-canonical public input", what);
  LightCircuit lc;
  lc.cap = s->special[b.kind].consts.cap;
  std::memcpy(lc.digest, s->special[b.kind].digest, 32);
  int r = rec_verify(s->rec_cfg, lc, b);
  if (r) {
    const std::string why = bp_last_error();
    return fail(BP_ERR_VERIFY, "%s does not verify: %s", what, why.c_str());
  }
  return BP_OK;
}

int verify_child(const bp_state* s, const Box& b, const char* what, const uint8_t* bytes, size_t len) {
  return produced_here(s, bytes, len) ? BP_OK : verify_foreign(s, b, what);
}

void root_after(const uint64_t root_before[4], uint64_t seed, uint64_t txn_number, uint64_t out[4]) {
  uint64_t in[6] = {root_before[0], root_before[1], root_before[2], root_before[3], gl::canon(seed), gl::canon(txn_number)};
  hash_no_pad_host(in, 6, out);
}

}  // namespace

extern "C" {

void bp_tune_rec_batch(int n) { g_rec_batch.store(n < 1 ? 1 : (n > (int)MAX_BATCH ? MAX_BATCH : (uint32_t)n)); }
void bp_tune_side_lanes(int n) { g_side_lanes.store(n < 0 ? 0 : n); }
void bp_tune_witness_threads(int n) { g_witness_threads.store(n < 1 ? 1 : (n > 16 ? 16 : n)); }

void bp_config_default(bp_config* c) {
  // constants.rs:6-18, positional order of prover_state.rs:85-93
  static const uint32_t lo[BP_NUM_TABLES] = {16, 9, 12, 14, 9, 12, 17}, hi[BP_NUM_TABLES] = {28, 28, 28, 25, 25, 28, 30};
  std::memset(c, 0, sizeof(*c));
  for (int t = 0; t < BP_NUM_TABLES; t++) { c->table_log_lo[t] = lo[t]; c->table_log_hi[t] = hi[t]; }
  c->stark_rate_bits = 1; c->stark_cap_height = 4; c->stark_num_queries = 84; c->stark_pow_bits = 16;
  c->arity_bits = 4; c->final_poly_bits = 5;
  // recursion-shaped proofs: proofs of the PLONK-shaped circuit (AIR 8), 135 wires, 85 preprocessed constant columns
  c->rec_log_n = 13; c->rec_n_cols = 135; c->rec_n_const = 85; c->rec_rate_bits = 3; c->rec_num_queries = 28;
  c->rec_pow_bits = 16;
  c->shrink_depth = 3;
  c->rec_air_id = 8;
  c->device = 0; c->n_workers = 4; c->arena_bytes = (uint64_t)6 << 30;
}

int bp_state_build(const bp_config* cfg, bp_state** out) try {
  if (!cfg || !out) return fail(BP_ERR_INVALID_INPUT, "bp_state_build: null argument");
  *out = nullptr;
  const StarkCfg rc = rec_cfg_of(*cfg);
  int r = check_cfg(rc);
  if (r) return r;
  if (cfg->shrink_depth < 1)
    return fail(BP_ERR_INVALID_INPUT, "shrink_depth must be at least 1: the root circuit walks Merkle paths of the recursion shape's depth, "
                "so its children are recursion-shaped proofs, not the tables' STARK proofs");
  uint32_t n_circ = 0;
  for (int t = 0; t < BP_NUM_TABLES; t++) {
    if (cfg->table_log_lo[t] >= cfg->table_log_hi[t] || cfg->table_log_hi[t] > 31)
      return fail(BP_ERR_INVALID_INPUT, "empty or invalid range for table %s", TABLE_NAMES[t]);
    StarkCfg tc = table_cfg_of(*cfg, cfg->table_log_lo[t], 8);
    if ((r = check_cfg(tc))) return r;
    n_circ += cfg->table_log_hi[t] - cfg->table_log_lo[t];
  }
  if (cfg->n_workers == 0 || cfg->n_workers > 64) return fail(BP_ERR_INVALID_INPUT, "n_workers out of range");
  int n_dev = bp_device_count();
  if (n_dev <= 0) return fail(BP_ERR_DEVICE, "no HIP device visible: the hot path has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= n_dev) return fail(BP_ERR_DEVICE, "device %d not present", cfg->device);
  // prover threads must sleep, not spin, while they wait (capi.cpp); a refusal (context configured by
  // another library already) is not an error: the prover then works with spinning waits
  (void)bp_use_blocking_sync(cfg->device);
  std::unique_ptr<bp_state> s(new bp_state());
  s->cfg = *cfg;
  s->rec_cfg = rc;
  // persistent memory: per circuit K*n values + K*n coeffs + K*m LDE + digests
  const uint64_t N = (uint64_t)1 << rc.log_n, M = N << rc.rate_bits;
  const size_t per = ((size_t)rc.n_const * (2 * N + M) + 2 * M * 4 + 4096) * 8;
  const size_t circuits_bytes = per * (n_circ + 3 + BP_NUM_TABLES) + (64u << 20);
  {
    // size the whole state against the device BEFORE the first allocation: a late hipMalloc failure would name one
    // arena, not the configuration that does not fit
    (void)hipSetDevice(cfg->device);
    size_t free_b = 0, total_b = 0;
    BPG_HIP(hipMemGetInfo(&free_b, &total_b));
    const double need = (double)circuits_bytes + (double)cfg->n_workers * (double)cfg->arena_bytes;
    if (need > (double)free_b)
      return fail(BP_ERR_DEVICE,
                  "prover state does not fit device %d: %u preprocessed circuits %.1f GiB + %u prover arenas x %.1f GiB = "
                  "%.1f GiB, %.1f GiB free of %.1f GiB (lower n_workers or arena_bytes, or narrow the table ranges)",
                  cfg->device, n_circ + 3 + BP_NUM_TABLES, circuits_bytes / 1073741824.0, cfg->n_workers,
                  cfg->arena_bytes / 1073741824.0, need / 1073741824.0, free_b / 1073741824.0, total_b / 1073741824.0);
  }
  // from here on device memory is owned by *s: release it on every early return
  struct Unbuild {
    std::unique_ptr<bp_state>& s;
    ~Unbuild() {
      if (!s) return;
      for (auto& w : s->workers) w->destroy();
      s->builder.destroy();
    }
  } unbuild{s};
  if ((r = s->builder.init(cfg->device, circuits_bytes))) return r;
  s->table_circuits.resize(n_circ);
  uint32_t idx = 0;
  for (int t = 0; t < BP_NUM_TABLES; t++) {
    s->table_offset[t] = idx;
    for (uint32_t d = cfg->table_log_lo[t]; d < cfg->table_log_hi[t]; d++, idx++)
      // (digest, table, level) and the child's (leaf digest, cap entry): the child of level 0 is the table's STARK proof,
      // whose trace tree has d + rate - cap levels below its cap
      if ((r = build_circuit(s->builder, rc, circuit_seed(t, d),
                             air::plonk::Layout{CHAIN_PATH_PI0 + 8, 1, d + cfg->stark_rate_bits - cfg->stark_cap_height, CHAIN_PATH_PI0},
                             &s->table_circuits[idx]))) return r;
  }
  // Every recursion circuit walks one Merkle path per child: root (seven chains), aggregation (two children: their
  // paths' words follow the digests and flags), block (the aggregation child's), a table's shrink circuit (the level below)
  const uint32_t depth = rc.log_n + rc.rate_bits - rc.cap_height;
  // Every circuit whose children are recursion-shaped proofs (all but level 0 of a chain, whose child is a table's STARK
  // proof of up to 2432 columns) also HASHES the row each child opens (its leaf: the child's n_cols trace values) before
  // it walks up from it -- merkle_proofs::verify_merkle_proof_to_cap whole -- where the circuit has the rows for it (a
  // 2^6-row test circuit has them for one child, not for seven: both sides take the same decision from the shape alone)
  auto with_leaf = [&](air::plonk::Layout lay) {
    air::plonk::Layout l = lay;
    l.leaf_len = rc.n_cols;
    return (rc.air_id == air::PLONK && air::plonk::layout_ok(l, 1u << rc.log_n)) ? l : lay;
  };
  for (int t = 0; t < BP_NUM_TABLES; t++)
    if ((r = build_circuit(s->builder, rc, circuit_seed(t, SHRINK_SEED_DEGREE), with_leaf(air::plonk::Layout{CHAIN_PATH_PI0 + 8, 1, depth, CHAIN_PATH_PI0}),
                           &s->shrink_circuits[t]))) return r;
  air::plonk::Layout special[3] = {with_leaf({ROOT_PATH_PI0 + 8 * BP_NUM_TABLES + BP_PV_WORDS, BP_NUM_TABLES, depth, ROOT_PATH_PI0}),
                                   with_leaf({AGG_PATH_PI0 + 2 * 8 + BP_PV_WORDS, 2, depth, AGG_PATH_PI0}),
                                   with_leaf({BLOCK_PATH_PI0 + 8 + BP_PV_WORDS, 1, depth, BLOCK_PATH_PI0})};
  for (uint32_t k = 0; k < 3; k++)
    if ((r = build_circuit(s->builder, rc, circuit_seed(CIRCUIT_ROOT + k, 0), special[k], &s->special[k]))) return r;
  BPG_HIP(hipStreamSynchronize(s->builder.stream));
  for (uint32_t i = 0; i < cfg->n_workers; i++) {
    std::unique_ptr<Worker> w(new Worker());
    if ((r = w->init(cfg->device, cfg->arena_bytes))) { w->destroy(); return r; }
    s->idle.push_back(w.get());
    s->workers.push_back(std::move(w));
  }
  {
    // One prover = one HIP stream, and ROCm multiplexes a process's streams over GPU_MAX_HW_QUEUES hardware queues
    // (4 when unset), read once when the HIP runtime starts: provers beyond that number share a queue and run one
    // after the other -- the state works, at a fraction of its rate.  The library cannot raise the limit after the
    // runtime has started, so it says so.
    const char* env = std::getenv("GPU_MAX_HW_QUEUES");
    const long queues = env && *env ? std::strtol(env, nullptr, 10) : 4;
    if (queues < (long)cfg->n_workers) {
      char buf[512];
      std::snprintf(buf, sizeof(buf),
                    "GPU_MAX_HW_QUEUES is %s%ld but this state has %u prover streams: HIP maps a process's streams onto that many "
                    "hardware queues, so concurrent bp_generate_* calls beyond it serialise (measured: 16 streams on 4 queues run at "
                    "about half the rate).  Set GPU_MAX_HW_QUEUES >= n_workers (bench.py uses 32) in the environment BEFORE the process "
                    "makes its first HIP call.",
                    env && *env ? "" : "unset = ", queues, cfg->n_workers);
      s->warnings = buf;
    }
    if (bp_host_wait_mode(cfg->device) == 2) {
      if (!s->warnings.empty()) s->warnings += "\n";
      s->warnings += "The process had used this device before the library's first call, so the device's host-wait mode was left alone "
                     "(bp_use_blocking_sync); the library's prover threads wait with their own poll-and-sleep loop instead of the "
                     "runtime's spinning wait.  Nothing to do; call bp_use_blocking_sync(device) first thing for interrupt-driven waits.";
    }
  }
  *out = s.release();
  return BP_OK;
}
BPG_ABI_CATCH("bp_state_build")

const char* bp_state_warnings(const bp_state* s) { return s ? s->warnings.c_str() : ""; }
int bp_state_config(const bp_state* s, bp_config* out) {
  if (!s || !out) return fail(BP_ERR_INVALID_INPUT, "bp_state_config: null argument");
  *out = s->cfg;
  return BP_OK;
}

void bp_state_free(bp_state* s) {
  if (!s) return;
  (void)hipSetDevice(s->cfg.device);
  for (auto& w : s->workers) w->destroy();
  s->builder.destroy();
  delete s;
}

uint64_t bp_state_device_bytes(const bp_state* s) {
  if (!s) return 0;
  return s->builder.arena.capacity() + (uint64_t)s->workers.size() * s->cfg.arena_bytes;
}

int bp_ir_encode(uint64_t block_number, uint64_t txn_number_before, uint64_t gas_used_before,
                 uint64_t gas_used_after, const uint64_t state_root_before[4], uint64_t seed,
                 const uint32_t table_log_n[BP_NUM_TABLES], const uint32_t table_width[BP_NUM_TABLES],
                 uint64_t o[BP_IR_WORDS]) {
  if (!state_root_before || !table_log_n || !table_width || !o) return fail(BP_ERR_INVALID_INPUT, "bp_ir_encode: null argument");
  o[0] = IR_MAGIC; o[1] = 1; o[2] = block_number; o[3] = txn_number_before; o[4] = gas_used_before; o[5] = gas_used_after;
  for (int i = 0; i < 4; i++) {
    if (state_root_before[i] >= gl::P) return fail(BP_ERR_INVALID_INPUT, "state root word is not a canonical field element");
    o[6 + i] = state_root_before[i];
  }
  o[10] = seed;
  for (int t = 0; t < BP_NUM_TABLES; t++) { o[11 + t] = table_log_n[t]; o[18 + t] = table_width[t]; }
  return BP_OK;
}

int bp_ir_set_keccak_air(uint64_t ir[BP_IR_WORDS], int on) {
  if (!ir || ir[0] != IR_MAGIC) return fail(BP_ERR_INVALID_INPUT, "bp_ir_set_keccak_air: not an IR");
  if (on && ir[18 + 3] != air::keccak::N_COLS)
    return fail(BP_ERR_INVALID_INPUT, "the Keccak-f AIR has %u columns: the IR gives table keccak %llu", air::keccak::N_COLS,
                (unsigned long long)ir[18 + 3]);
  ir[1] = (ir[1] & ~(uint64_t)0x100) | (on ? 0x100 : 0);
  return BP_OK;
}

int bp_ir_set_logic_air(uint64_t ir[BP_IR_WORDS], int on) {
  if (!ir || ir[0] != IR_MAGIC) return fail(BP_ERR_INVALID_INPUT, "bp_ir_set_logic_air: not an IR");
  if (on && ir[18 + 5] != air::logic::N_COLS)
    return fail(BP_ERR_INVALID_INPUT, "the logic AIR has %u columns: the IR gives table logic %llu", air::logic::N_COLS,
                (unsigned long long)ir[18 + 5]);
  ir[1] = (ir[1] & ~(uint64_t)0x200) | (on ? 0x200 : 0);
  return BP_OK;
}

int bp_ir_set_memory_air(uint64_t ir[BP_IR_WORDS], int on) {
  if (!ir || ir[0] != IR_MAGIC) return fail(BP_ERR_INVALID_INPUT, "bp_ir_set_memory_air: not an IR");
  if (on && ir[18 + 6] != air::memory::N_COLS)
    return fail(BP_ERR_INVALID_INPUT, "the memory AIR has %u columns: the IR gives table memory %llu", air::memory::N_COLS,
                (unsigned long long)ir[18 + 6]);
  ir[1] = (ir[1] & ~(uint64_t)0x400) | (on ? 0x400 : 0);
  return BP_OK;
}

int bp_ir_set_arithmetic_air(uint64_t ir[BP_IR_WORDS], int on) {
  if (!ir || ir[0] != IR_MAGIC) return fail(BP_ERR_INVALID_INPUT, "bp_ir_set_arithmetic_air: not an IR");
  if (on && ir[18 + 0] != air::arithmetic::N_COLS)
    return fail(BP_ERR_INVALID_INPUT, "the arithmetic AIR has %u columns: the IR gives table arithmetic %llu",
                air::arithmetic::N_COLS, (unsigned long long)ir[18 + 0]);
  ir[1] = (ir[1] & ~(uint64_t)0x800) | (on ? 0x800 : 0);
  return BP_OK;
}

int bp_ir_set_arithmetic_mul_air(uint64_t ir[BP_IR_WORDS], int on) {
  if (!ir || ir[0] != IR_MAGIC) return fail(BP_ERR_INVALID_INPUT, "bp_ir_set_arithmetic_mul_air: not an IR");
  if (on && ir[18 + 0] != air::arithmetic_mul::N_COLS)
    return fail(BP_ERR_INVALID_INPUT, "the multiplication AIR has %u columns: the IR gives table arithmetic %llu",
                air::arithmetic_mul::N_COLS, (unsigned long long)ir[18 + 0]);
  if (on && (ir[1] & 0x800)) return fail(BP_ERR_INVALID_INPUT, "the arithmetic table is proven by ONE AIR: clear bp_ir_set_arithmetic_air first");
  ir[1] = (ir[1] & ~(uint64_t)0x4000) | (on ? 0x4000 : 0);
  return BP_OK;
}

int bp_ir_set_byte_packing_air(uint64_t ir[BP_IR_WORDS], int on) {
  if (!ir || ir[0] != IR_MAGIC) return fail(BP_ERR_INVALID_INPUT, "bp_ir_set_byte_packing_air: not an IR");
  if (on && ir[18 + 1] != air::byte_packing::N_COLS)
    return fail(BP_ERR_INVALID_INPUT, "the byte-packing AIR has %u columns: the IR gives table byte_packing %llu",
                air::byte_packing::N_COLS, (unsigned long long)ir[18 + 1]);
  ir[1] = (ir[1] & ~(uint64_t)0x1000) | (on ? 0x1000 : 0);
  return BP_OK;
}

int bp_ir_set_keccak_sponge_air(uint64_t ir[BP_IR_WORDS], int on) {
  if (!ir || ir[0] != IR_MAGIC) return fail(BP_ERR_INVALID_INPUT, "bp_ir_set_keccak_sponge_air: not an IR");
  if (on && ir[18 + 4] != air::keccak_sponge::N_COLS)
    return fail(BP_ERR_INVALID_INPUT, "the Keccak sponge AIR has %u columns: the IR gives table keccak_sponge %llu",
                air::keccak_sponge::N_COLS, (unsigned long long)ir[18 + 4]);
  ir[1] = (ir[1] & ~(uint64_t)0x2000) | (on ? 0x2000 : 0);
  return BP_OK;
}

int bp_ir_encode_dummy(uint64_t block_number, uint64_t txn_number, uint64_t gas_used, const uint64_t state_root[4],
                       uint64_t seed, const uint32_t table_log_n[BP_NUM_TABLES], const uint32_t table_width[BP_NUM_TABLES],
                       uint64_t o[BP_IR_WORDS]) {
  int rc = bp_ir_encode(block_number, txn_number, gas_used, gas_used, state_root, seed, table_log_n, table_width, o);
  if (rc == BP_OK) o[1] = 2;
  return rc;
}

// root_after of one txn (the synthetic "state transition"): hash_no_pad(root_before, seed, txn_number).
// Host-side, like the decoder that chains GenerationInputs in the reference (decoding.rs:106-154).
int bp_state_root_after(const uint64_t root_before[4], uint64_t seed, uint64_t txn_number, uint64_t out[4]) {
  if (!root_before || !out) return fail(BP_ERR_INVALID_INPUT, "bp_state_root_after: null argument");
  for (int i = 0; i < 4; i++) if (root_before[i] >= gl::P) return fail(BP_ERR_INVALID_INPUT, "non-canonical state root");
  root_after(root_before, seed, txn_number, out);
  return BP_OK;
}

int bp_proof_public_values(const uint8_t* proof, size_t len, uint64_t pv_out[BP_PV_WORDS], int* kind_out) try {
  if (!proof || len % 8 || len < (BOX_HDR + BP_PV_WORDS) * 8) return fail(BP_ERR_INVALID_INPUT, "proof: truncated");
  const uint64_t* w = reinterpret_cast<const uint64_t*>(proof);
  if (w[0] != PROOF_BOX_MAGIC || w[1] > 2 || w[2] < BP_PV_WORDS || w[2] > air::plonk::MAX_PI || len / 8 < BOX_HDR + w[2])
    return fail(BP_ERR_INVALID_INPUT, "proof: bad header");
  if (pv_out) std::memcpy(pv_out, w + BOX_HDR + w[2] - BP_PV_WORDS, BP_PV_WORDS * 8);
  if (kind_out) *kind_out = (int)w[1];
  return BP_OK;
}
BPG_ABI_CATCH("bp_proof_public_values")

// Witness data given by the caller instead of drawn from the seed, per table (bp_txn_witness): in[t] nullable, n[t]
// items of WITNESS_WORDS[t] words; the table must carry its AIR flag.  The rest of the table is padding: permutations
// of the all-zero state (as upstream pads its Keccak table), rows without an operation, and for the memory log reads
// of the last address at later and later times (a memory that is left alone).
struct TxnWitness {
  const uint64_t* in[BP_NUM_TABLES] = {};
  size_t n[BP_NUM_TABLES] = {};
};
static const uint32_t WITNESS_WORDS[BP_NUM_TABLES] = {9, 6, 0, 25, 44, 9, 11};  // arithmetic, byte packing, -, keccak, sponge, logic, memory
static const uint32_t WITNESS_AIR[BP_NUM_TABLES] = {air::ARITHMETIC, air::BYTE_PACKING, ~0u, air::KECCAK_F, air::KECCAK_SPONGE, air::LOGIC, air::MEMORY};
// items a table of N rows holds: one permutation per 24 rows for Keccak (the last one may be cut), else one per row
static size_t witness_capacity(int t, uint64_t N) { return t == 3 ? (size_t)((N + 23) / 24) : (size_t)N; }
// the table's full input array (capacity x words), the caller's items first, then padding
static void fill_table_inputs(int t, uint64_t N, const uint64_t* in, size_t n, uint64_t* dst) {
  const size_t cap = witness_capacity(t, N), wds = WITNESS_WORDS[t];
  std::memcpy(dst, in, n * wds * 8);
  std::memset(dst + n * wds, 0, (cap - n) * wds * 8);
  if (t == 6) {  // memory: keep reading the last cell (a first-row read of zero memory when the log is empty)
    uint64_t last[11] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (n) std::memcpy(last, in + (n - 1) * 11, sizeof(last));
    last[0] = 1;
    for (size_t i = n; i < cap; i++) {
      last[2] += 1;
      std::memcpy(dst + i * 11, last, sizeof(last));
    }
  }
}
// What upstream's `prove` returns before the recursion starts (AllProof: the seven table proofs, the lookup
// challenges, the public values) -- kept by bp_generate_txn_table_proofs, digested by bp_generate_txn_proof.
struct TableProofs {
  StarkCfg tcfg[BP_NUM_TABLES];
  std::vector<uint64_t> pv;
  Ctl ctl;
  std::vector<uint64_t> proof[BP_NUM_TABLES];
  uint64_t first_leaf[BP_NUM_TABLES][4];  // per table proof: the digest of the trace leaf its first query opens (stark_prove)
};
static int parse_ir(const bp_config& cfg, const uint64_t* I, const TxnWitness* wit, StarkCfg tcfg[BP_NUM_TABLES],
                    std::vector<uint64_t>* pv_out) {
  // version 1: a transaction; version 2: a dummy entry (decoding.rs:484-520): txn number, gas and state
  // root do not advance, the same tables are proven
  // flags above the version byte: 0x100 = the Keccak table (index 3, prover_state.rs:85-93) is proven with the
  // Keccak-f AIR (air.hpp, AIR 1) instead of the synthetic one: 2431 columns, witness drawn from the seed;
  // 0x200 = the logic table (index 5) is proven with the logic AIR (AIR 2): 523 columns, operations drawn from the seed;
  // 0x400 = the memory table (index 6) with the memory AIR (AIR 3): 45 columns, a sorted log drawn from the seed;
  // 0x800 = the arithmetic table (index 0) with the arithmetic AIR (AIR 4): 309 columns;
  // 0x1000 = the byte-packing table (index 1) with the byte-packing AIR (AIR 5): 299 columns;
  // 0x2000 = the Keccak sponge table (index 4) with the Keccak sponge AIR (AIR 6): 2414 columns;
  // 0x4000 = the arithmetic table (index 0) with the MULTIPLICATION AIR (AIR 7, the multiplicative half of upstream's
  //          arithmetic table: 1217 columns) instead of AIR 4 -- one or the other
  const uint64_t ver = I[1] & 0xFF, flags = I[1] >> 8;
  if (I[0] != IR_MAGIC || (ver != 1 && ver != 2) || flags > 127) return fail(BP_ERR_INVALID_INPUT, "IR: bad magic/version");
  if ((flags & 8) && (flags & 64)) return fail(BP_ERR_INVALID_INPUT, "IR: the arithmetic table is proven by ONE AIR (flags 0x800 and 0x4000 are both set)");
  const bool dummy = ver == 2;
  // table index -> the AIR its flag selects (bit t' of flags; WITNESS_AIR lists them by table)
  static const uint32_t FLAG_OF_TABLE[BP_NUM_TABLES] = {8, 16, 0, 1, 32, 2, 4};
  if (wit) {
    for (int t = 0; t < BP_NUM_TABLES; t++) {
      if (!wit->in[t]) continue;
      if (!(flags & (FLAG_OF_TABLE[t] | (t == 0 ? 64u : 0u))))
        return fail(BP_ERR_INVALID_INPUT, "witness data for table %s needs an IR whose %s table is proven with its AIR (bp_ir_set_*_air)",
                    TABLE_NAMES[t], TABLE_NAMES[t]);
      if (I[11 + t] < 40 && wit->n[t] > witness_capacity(t, (uint64_t)1 << I[11 + t]))
        return fail(BP_ERR_RANGE, "%zu witness items do not fit table %s of 2^%llu rows%s", wit->n[t], TABLE_NAMES[t],
                    (unsigned long long)I[11 + t], t == 3 ? " (24 rows per Keccak permutation)" : "");
    }
  }
  if (I[5] < I[4]) return fail(BP_ERR_INVALID_INPUT, "IR: gas_used_after < gas_used_before");
  if (dummy && I[5] != I[4]) return fail(BP_ERR_INVALID_INPUT, "IR: a dummy entry must not use gas (decoding.rs:503-506)");
  for (int t = 0; t < BP_NUM_TABLES; t++) {
    const uint64_t ln = I[11 + t], wd = I[18 + t];
    if (ln < cfg.table_log_lo[t] || ln >= cfg.table_log_hi[t])
      return fail(BP_ERR_RANGE, "table %s needs 2^%llu rows, outside the configured range %u..%u", TABLE_NAMES[t],
                  (unsigned long long)ln, cfg.table_log_lo[t], cfg.table_log_hi[t]);
    if (wd > 65536) return fail(BP_ERR_INVALID_INPUT, "table %s: width out of range", TABLE_NAMES[t]);
    tcfg[t] = table_cfg_of(cfg, (uint32_t)ln, (uint32_t)wd);
    if (flags & FLAG_OF_TABLE[t]) tcfg[t].air_id = WITNESS_AIR[t];  // check_cfg insists on the AIR's own width
    if (t == 0 && (flags & 64)) tcfg[t].air_id = air::ARITHMETIC_MUL;
    int r = check_cfg(tcfg[t]);
    if (r) return r;
  }
  for (int i = 0; i < 4; i++) if (I[6 + i] >= gl::P) return fail(BP_ERR_INVALID_INPUT, "IR: non-canonical state root");
  // PublicValues
  std::vector<uint64_t>& pv = *pv_out;
  pv.assign(BP_PV_WORDS, 0);
  pv[0] = I[3]; pv[1] = I[3] + (dummy ? 0 : 1); pv[2] = I[4]; pv[3] = I[5];
  std::memcpy(&pv[4], I + 6, 32);
  if (dummy) std::memcpy(&pv[8], I + 6, 32);
  else root_after(I + 6, I[10], I[3], &pv[8]);
  pv[12] = I[2];
  for (auto& v : pv) v = gl::canon(v);
  return BP_OK;
}

// The cross-table lookups of a transaction's table proofs (air::ctl::pairs): for both challenge sets the first-row
// value of the looking running product equals that of the looked one.  Only pairs whose two tables are proven with
// their AIRs exist (a synthetic table has nothing to look up).  Shared by the prover (which refuses to go on with
// tables that do not form one statement: upstream's root circuit checks this in-circuit) and bp_verify_txn_table_proofs.
static int check_lookups(const StarkCfg tcfg[BP_NUM_TABLES], const std::vector<uint64_t> proof[BP_NUM_TABLES]) {
  const air::ctl::Pair* P = air::ctl::pairs();
  for (uint32_t i = 0; i < air::ctl::N_PAIRS; i++) {
    const air::ctl::Pair& p = P[i];
    if (tcfg[p.looking_table].air_id != p.looking_air || tcfg[p.looked_table].air_id != p.looked_air) continue;
    const ProofLayout La = proof_layout(tcfg[p.looking_table]), Lb = proof_layout(tcfg[p.looked_table]);
    for (uint32_t c = 0; c < 2; c++) {
      // the first-row values of the looking side's product columns, multiplied together, against the looked side's
      gl::Ext a = gl::ext(1);
      for (uint32_t m = 0; m < p.n_looking; m++) {
        const uint64_t* v = proof[p.looking_table].data() + La.open_first + 2 * (p.looking_col + p.stride * m + c);
        a = gl::mul(a, gl::Ext{v[0], v[1]});
      }
      const uint64_t* b = proof[p.looked_table].data() + Lb.open_first + 2 * (p.looked_col + c);
      if (a.c0 != b[0] || a.c1 != b[1])
        return fail(BP_ERR_VERIFY, "cross-table lookup %s does not hold (challenge set %u): the %s table asks for tuples the %s table "
                    "does not expose", p.name, c, TABLE_NAMES[p.looking_table], TABLE_NAMES[p.looked_table]);
    }
  }
  return BP_OK;
}

// generate_traces + the seven table proofs on one transcript (plonky2_evm `prove`), on the leased worker
static int prove_tables(const bp_state* s, Worker& w, const uint64_t* I, const TxnWitness* wit, TableProofs* tp) {
  StarkCfg* tcfg = tp->tcfg;
  int r;
  // generate_traces + trace commitments for all tables, then the shared transcript prologue
  uint64_t* d_trace[BP_NUM_TABLES];
  Committed trace[BP_NUM_TABLES];
  Challenger ch;
  auto given = [&](int t) { return wit && wit->in[t] && (tcfg[t].air_id == WITNESS_AIR[t] || (t == 0 && tcfg[t].air_id == air::ARITHMETIC_MUL)); };
  // Two seeded tables that a lookup ties together are ONE statement: the seeded sponge table asks for no more
  // permutations than the Keccak-f table holds in full, and the seeded Keccak-f table's first permutations are the
  // ones the sponge rows ask for (air::ctl, keccak_sponge -> keccak_f).  Tables given by the caller are taken as they are.
  const bool lookup_kf = tcfg[3].air_id == air::KECCAK_F && tcfg[4].air_id == air::KECCAK_SPONGE;
  // keccak_sponge -> logic: the XOR of every absorbed block with the rate is five operations of the logic table, whose
  // first rows are then derived from the sponge table's trace (five per covered sponge row; the caller's or seeded
  // operations follow them); the seeded sponge table absorbs no more blocks than the logic table can hold
  const bool lookup_sl = tcfg[4].air_id == air::KECCAK_SPONGE && tcfg[5].air_id == air::LOGIC;
  const uint32_t logic_covered = lookup_sl ? (uint32_t)std::min<uint64_t>((uint64_t)1 << tcfg[4].log_n, ((uint64_t)1 << tcfg[5].log_n) / 5) : 0;
  const uint32_t sponge_row_limit = std::min<uint32_t>(lookup_kf ? (uint32_t)(((uint64_t)1 << tcfg[3].log_n) / 24) : ~0u,
                                                       lookup_sl ? logic_covered : ~0u);
  // byte_packing -> memory: the memory table that is not given by the caller is the log of the byte-packing table's
  // words (two operations per packing row); it must be tall enough to hold them
  const bool lookup_bm = tcfg[1].air_id == air::BYTE_PACKING && tcfg[6].air_id == air::MEMORY;
  if (lookup_bm && given(1) && !given(6))
    return fail(BP_ERR_INVALID_INPUT, "byte-packing sequences are given but the memory log is not: the memory table is looked up by them "
                "(byte_packing -> memory) and cannot be drawn from the seed");
  // the mirror cases: a LOOKED table given by the caller while its looking table is drawn from the seed cannot be one
  // statement with it either (the seeded sponge rows ask for permutations of their own; the seeded packing rows move
  // words of their own) -- refused here, before seven table proofs are made and check_lookups blames the tables
  if (lookup_kf && given(3) && !given(4))
    return fail(BP_ERR_INVALID_INPUT, "Keccak-f permutations are given but the sponge rows are not: with both tables proven by their "
                "AIRs the sponge table looks the permutations up (keccak_sponge -> keccak_f); give the sponge rows too "
                "(bp_txn_witness.sponge_rows, bp_keccak256_sponge_rows) or clear the sponge table's AIR flag");
  if (lookup_bm && given(6) && !given(1))
    return fail(BP_ERR_INVALID_INPUT, "the memory log is given but the byte-packing sequences are not: the seeded byte-packing table "
                "looks up operations of its own (byte_packing -> memory); give the sequences too or clear one of the two AIR flags");
  if (lookup_bm && !given(6) && tcfg[6].log_n < tcfg[1].log_n + 1)
    return fail(BP_ERR_INVALID_INPUT, "the memory table (2^%u rows) cannot hold the operations of the byte-packing table (2^%u rows): "
                "two per row", tcfg[6].log_n, tcfg[1].log_n);
  static const int GEN_ORDER[BP_NUM_TABLES] = {4, 0, 1, 2, 3, 5, 6};  // a looking table before the table it looks up (sponge before Keccak-f, byte packing before memory)
  for (int t = 0; t < BP_NUM_TABLES; t++) {
    const uint64_t N = (uint64_t)1 << tcfg[t].log_n;
    d_trace[t] = w.arena.alloc_words((size_t)tcfg[t].n_cols * N);
    if (!d_trace[t]) return fail(BP_ERR_DEVICE, "device arena exhausted (%zu MiB) for table %s", w.arena.capacity() >> 20, TABLE_NAMES[t]);
  }
  for (int gi = 0; gi < BP_NUM_TABLES; gi++) {
    const int t = GEN_ORDER[gi];
    const uint64_t N = (uint64_t)1 << tcfg[t].log_n, seed = I[10] ^ splitmix64(t + 1);
    const size_t mark = w.arena.mark();
    uint64_t* d_in = nullptr;
    if (t == 5 && lookup_sl) {
      // [five operations per covered sponge row][the caller's operations, or seeded ones]
      const size_t n_given = given(5) ? wit->n[5] : 0;
      if (n_given > N - 5ull * logic_covered)
        return fail(BP_ERR_INVALID_INPUT, "the logic table (2^%u rows) holds the sponge table's %u XORs first: room for %llu operations, %zu given",
                    tcfg[5].log_n, 5 * logic_covered, (unsigned long long)(N - 5ull * logic_covered), n_given);
      d_in = w.arena.alloc_words((size_t)N * 9 + n_given * 9);
      if (!d_in) return fail(BP_ERR_DEVICE, "device arena exhausted for the logic table's operations");
      uint64_t* d_given = nullptr;
      if (n_given) {
        if (n_given * 9 > w.pinned_words) return fail(BP_ERR_UNSUPPORTED, "too many logic operations for the input staging buffer");
        std::memcpy(w.pinned, wit->in[5], n_given * 9 * 8);
        d_given = d_in + (size_t)N * 9;
        BPG_HIP(hipMemcpyAsync(d_given, w.pinned, n_given * 9 * 8, hipMemcpyHostToDevice, w.stream));
      }
      if ((r = launch_logic_inputs_from_sponge(d_trace[4], tcfg[4].log_n, logic_covered, given(5) ? d_given : nullptr, (uint32_t)n_given,
                                               d_in, (uint32_t)N, seed, w.stream))) return r;
      if (given(5) && !d_given) {  // an empty list was given: padding operations, not seeded ones
        BPG_HIP(hipMemsetAsync(d_in + (size_t)5 * logic_covered * 9, 0, ((size_t)N - 5 * logic_covered) * 9 * 8, w.stream));
      }
    } else if (given(t)) {
      // the caller's items, then padding up to the table's height, staged through the pinned buffer
      const size_t words = witness_capacity(t, N) * WITNESS_WORDS[t];
      d_in = w.arena.alloc_words(words);
      if (!d_in) return fail(BP_ERR_DEVICE, "device arena exhausted for the witness data of table %s", TABLE_NAMES[t]);
      if (words > w.pinned_words) return fail(BP_ERR_UNSUPPORTED, "table %s too tall for the input staging buffer", TABLE_NAMES[t]);
      fill_table_inputs(t, N, wit->in[t], wit->n[t], w.pinned);
      BPG_HIP(hipMemcpyAsync(d_in, w.pinned, words * 8, hipMemcpyHostToDevice, w.stream));
    } else if (t == 3 && lookup_kf) {
      const uint32_t n_perms = (uint32_t)witness_capacity(3, N);
      d_in = w.arena.alloc_words((size_t)n_perms * 25);
      if (!d_in) return fail(BP_ERR_DEVICE, "device arena exhausted for the Keccak-f table's inputs");
      if ((r = launch_keccak_inputs_from_sponge(d_trace[4], tcfg[4].log_n, d_in, n_perms, seed, w.stream))) return r;
    } else if (t == 6 && lookup_bm) {
      d_in = w.arena.alloc_words((size_t)N * 11);
      if (!d_in) return fail(BP_ERR_DEVICE, "device arena exhausted for the memory table's log");
      if ((r = launch_memory_inputs_from_byte_packing(d_trace[1], tcfg[1].log_n, d_in, (uint32_t)N, w.stream))) return r;
    }
    switch (tcfg[t].air_id) {
      case air::KECCAK_F: r = launch_keccak_trace(d_trace[t], d_in, tcfg[t].log_n, d_in ? 0 : seed, w.stream); break;
      case air::LOGIC: r = launch_logic_trace(d_trace[t], d_in, tcfg[t].log_n, d_in ? 0 : seed, w.stream); break;
      case air::MEMORY: r = launch_memory_trace(d_trace[t], d_in, tcfg[t].log_n, d_in ? 0 : seed, w.stream); break;
      case air::ARITHMETIC: r = launch_arithmetic_trace(d_trace[t], d_in, tcfg[t].log_n, d_in ? 0 : seed, w.stream); break;
      case air::ARITHMETIC_MUL: r = launch_arithmetic_mul_trace(d_trace[t], d_in, tcfg[t].log_n, d_in ? 0 : seed, w.stream); break;
      case air::BYTE_PACKING: r = launch_byte_packing_trace(d_trace[t], d_in, tcfg[t].log_n, d_in ? 0 : seed, w.stream); break;
      case air::KECCAK_SPONGE:
        r = launch_keccak_sponge_trace(d_trace[t], d_in, tcfg[t].log_n, d_in ? 0 : seed, w.stream, sponge_row_limit);
        break;
      default: r = launch_synth_trace(d_trace[t], nullptr, tcfg[t].log_n, tcfg[t].n_cols, 0, 1, seed, w.stream); break;
    }
    if (r) return r;
    // the filter column of a looked table (air::ctl) is part of its TRACE: written here, committed with the trace, i.e.
    // before the lookup challenges are drawn.  The looking table's trace is there already (GEN_ORDER).
    if (t == 3 && lookup_kf) {  // the permutations the sponge table asks for: its flag columns, row p <-> permutation p
      const uint64_t N4 = (uint64_t)1 << tcfg[4].log_n;
      r = launch_lookup_filter(air::KECCAK_F, d_trace[3], tcfg[3].log_n, d_trace[4] + (size_t)air::keccak_sponge::COL_FULL * N4,
                               d_trace[4] + (size_t)air::keccak_sponge::COL_FINAL * N4, (uint32_t)N4, w.stream);
    } else if (t == 6 && lookup_bm) {  // the operations the byte-packing table looks up: its trace (address, timestamp per row)
      r = launch_lookup_filter(air::MEMORY, d_trace[6], tcfg[6].log_n, d_trace[1], nullptr, (uint32_t)1 << tcfg[1].log_n, w.stream);
    } else if (t == 5 && lookup_sl) {  // the XORs the sponge table asks for: rows 5 p + m of the rows p that absorb a block
      const uint64_t N4 = (uint64_t)1 << tcfg[4].log_n;
      r = launch_lookup_filter(air::LOGIC, d_trace[5], tcfg[5].log_n, d_trace[4] + (size_t)air::keccak_sponge::COL_FULL * N4,
                               d_trace[4] + (size_t)air::keccak_sponge::COL_FINAL * N4, logic_covered, w.stream);
    }
    if (r) return r;
    if (d_in) {  // the staging buffer and the input words are reused by the next table
      if ((r = w.wait())) return r;
      w.arena.release(mark);
    }
  }
  // The seven trace commitments do not depend on each other.  Under load they queue on this prover's stream like
  // everything else (the chip is full); a prover that is ALONE on the device -- a lone transaction, the last one of a
  // shard -- borrows the streams of up to three idle workers and the commitments overlap: the wide Keccak table's long
  // sponge chains no longer have the chip to themselves (largest first, each to the lane with the least work so far).
  std::vector<std::unique_ptr<SideLane>> sides;
  if (s && provers_active() <= g_side_lanes.load(std::memory_order_relaxed))
    for (int k = 0; k < 3; k++) {
      std::unique_ptr<SideLane> l = SideLane::try_acquire(s);
      if (!l) break;
      sides.push_back(std::move(l));
    }
  if (sides.empty()) {
    for (int t = 0; t < BP_NUM_TABLES; t++)
      if ((r = commit(w, d_trace[t], tcfg[t].n_cols, tcfg[t].log_n, tcfg[t].rate_bits, tcfg[t].cap_height, false, &trace[t]))) return r;
  } else {
    BPG_HIP(hipEventRecord(w.sync_event, w.stream));  // the witnesses are made on this prover's stream
    std::vector<Worker*> lane = {&w};
    for (auto& l : sides) {
      BPG_HIP(hipStreamWaitEvent(l->w->stream, w.sync_event, 0));
      lane.push_back(l->w);
    }
    std::vector<uint64_t> load(lane.size(), 0);
    std::vector<size_t> slots(lane.size(), 0);
    int order[BP_NUM_TABLES];
    for (int t = 0; t < BP_NUM_TABLES; t++) order[t] = t;
    auto cells = [&](int t) { return (uint64_t)tcfg[t].n_cols << tcfg[t].log_n; };
    std::sort(order, order + BP_NUM_TABLES, [&](int a, int b) { return cells(a) != cells(b) ? cells(a) > cells(b) : a < b; });
    PendingCommit pc[BP_NUM_TABLES];
    for (int oi = 0; oi < BP_NUM_TABLES; oi++) {
      const int t = order[oi];
      const size_t k = std::min_element(load.begin(), load.end()) - load.begin();
      if ((r = commit_launch(w, *lane[k], slots[k], d_trace[t], tcfg[t].n_cols, 1, tcfg[t].log_n, tcfg[t].rate_bits,
                             tcfg[t].cap_height, false, &pc[t])))
        return r;
      load[k] += cells(t);
      slots[k] += (size_t)4 << tcfg[t].cap_height;
    }
    for (int t = 0; t < BP_NUM_TABLES; t++)
      if ((r = commit_finish(pc[t], &trace[t]))) return r;
    sides.clear();  // the lanes are drained (commit_finish waited for each): back to their owners
  }
  for (int t = 0; t < BP_NUM_TABLES; t++) ch.observe(trace[t].cap.data(), trace[t].cap.size());
  ch.observe(tp->pv.data(), tp->pv.size());
  Ctl& ctl = tp->ctl;
  for (int i = 0; i < 4; i++) ctl.v[i] = ch.challenge();
  // table proofs: sequential, one transcript threaded through all of them (plonky2_evm prover)
  for (int t = 0; t < BP_NUM_TABLES; t++) {
    if (w.aborted()) return fail(BP_ERR_ABORTED, "aborted before table %s", TABLE_NAMES[t]);
    const size_t mark = w.arena.mark();
    Challenger before = ch;  // the transcript as the verifier of this table proof starts from it
    if ((r = stark_prove(w, tcfg[t], nullptr, trace[t], d_trace[t], ctl, ch, tp->proof[t], tp->first_leaf[t]))) return r;
    if (given(t)) {
      // The prover does not check a witness, and nothing downstream of this call verifies the table proofs (upstream's
      // root circuit would): data that came from the caller is therefore checked here, by the CPU verifier on the
      // proof just made -- a log that is not a memory, a block that is not padded, ... ends the call.
      if (stark_verify(tcfg[t], nullptr, ctl, before, tp->proof[t].data(), tp->proof[t].size()) != BP_OK) {
        const std::string why = bp_last_error();
        return fail(BP_ERR_VERIFY, "the witness data given for table %s does not satisfy its AIR: %s", TABLE_NAMES[t], why.c_str());
      }
    }
    w.arena.release(mark);
  }
  // the tables must be ONE statement: what the sponge table hashes is what the Keccak-f table permutes
  return check_lookups(tcfg, tp->proof);
}

static int txn_proof_impl(const bp_state* s, const uint8_t* ir, size_t ir_len, const volatile int32_t* abort_flag,
                          const volatile uint8_t* abort_flag_u8, uint8_t** out, size_t* out_len,
                          const TxnWitness* wit = nullptr, bool tables_only = false) {
  if (!s || !ir || !out || !out_len) return fail(BP_ERR_INVALID_INPUT, "bp_generate_txn_proof: null argument");
  if (ir_len != BP_IR_WORDS * 8) return fail(BP_ERR_INVALID_INPUT, "IR must be %d bytes", BP_IR_WORDS * 8);
  const uint64_t* I = reinterpret_cast<const uint64_t*>(ir);
  const bp_config& cfg = s->cfg;
  TableProofs tp;
  int r = parse_ir(s->cfg, I, wit, tp.tcfg, &tp.pv);
  if (r) return r;
  const StarkCfg* tcfg = tp.tcfg;
  const std::vector<uint64_t>& pv = tp.pv;

  (void)hipSetDevice(cfg.device);
  WorkerLease lease(s);
  Worker& w = *lease.w;
  w.abort_flag = abort_flag;
  w.abort_flag_u8 = abort_flag_u8;
  if (w.aborted()) return fail(BP_ERR_ABORTED, "aborted before start");
  if ((r = prove_tables(s, w, I, wit, &tp))) return r;
  if (tables_only) {
    // "BPGTABLS" | n_tables | public values | lookup challenges | per table: air_id, log_n, n_cols, n_words, proof words
    std::vector<uint64_t> o = {TABLES_MAGIC, BP_NUM_TABLES};
    o.insert(o.end(), pv.begin(), pv.end());
    o.insert(o.end(), tp.ctl.v, tp.ctl.v + 4);
    for (int t = 0; t < BP_NUM_TABLES; t++) {
      const uint64_t hdr[4] = {tcfg[t].air_id, tcfg[t].log_n, tcfg[t].n_cols, tp.proof[t].size()};
      o.insert(o.end(), hdr, hdr + 4);
      o.insert(o.end(), tp.proof[t].begin(), tp.proof[t].end());
    }
    uint64_t* buf = static_cast<uint64_t*>(std::malloc(o.size() * 8));
    if (!buf) return fail(BP_ERR_DEVICE, "host allocation failed");
    std::memcpy(buf, o.data(), o.size() * 8);
    *out = reinterpret_cast<uint8_t*>(buf);
    *out_len = o.size() * 8;
    return BP_OK;
  }
  // per child of a recursion circuit: its digest, and the Merkle path of its first trace opening (leaf digest, cap entry:
  // words of the parent's public-input list; position and siblings: witness of the parent's Merkle rows)
  uint64_t digest[BP_NUM_TABLES][4], leaf_cap[BP_NUM_TABLES][8];
  std::vector<PathWitness> child_path[BP_NUM_TABLES];
  for (int t = 0; t < BP_NUM_TABLES; t++) {
    proof_digest(tcfg[t], tp.proof[t].data(), digest[t]);
    child_path[t].resize(1);
    first_query_trace_path(tcfg[t], tp.proof[t].data(), leaf_cap[t], leaf_cap[t] + 4, &child_path[t][0], tp.first_leaf[t]);
    std::vector<uint64_t>().swap(tp.proof[t]);
  }
  std::vector<uint64_t> proof;
  if ((r = w.wait())) return r;
  w.arena.release(lease.mark);  // traces are dead; the chains below only need digests
  // per-table recursion-shaped chains (wrap + shrinks).  The seven chains do not depend on each other and their
  // proofs have one shape: level k of all seven is ONE batch proved in lock-step (rec_prove_batch) -- seven
  // transcripts stepped together, every kernel launch and host wait shared --, then level k + 1.  (Until round 4 each
  // chain was its own sequence of 6..16-column launches: 88 of a transaction's 118 LDE launches and most of its
  // 1,620 kernel launches.)  Level 0 is proved by the circuit of the table's height (its child is the table's STARK
  // proof), the levels above by the table's shrink circuit.
  {
    const Circuit* circ[BP_NUM_TABLES];
    std::vector<uint64_t> pis[BP_NUM_TABLES], chain_proof[BP_NUM_TABLES];
    for (uint32_t depth = 0; depth < cfg.shrink_depth; depth++) {
      if (w.aborted()) return fail(BP_ERR_ABORTED, "aborted in the recursion chains (level %u)", depth);
      for (int t = 0; t < BP_NUM_TABLES; t++) {
        circ[t] = depth == 0 ? &s->table_circuits[s->table_offset[t] + (tcfg[t].log_n - cfg.table_log_lo[t])] : &s->shrink_circuits[t];
        pis[t] = {digest[t][0], digest[t][1], digest[t][2], digest[t][3], (uint64_t)t, depth};
        pis[t].insert(pis[t].end(), leaf_cap[t], leaf_cap[t] + 8);
      }
      uint64_t first_leaf[BP_NUM_TABLES][4];
      if ((r = rec_prove_batch(w, s->rec_cfg, BP_NUM_TABLES, circ, pis, chain_proof, child_path, &first_leaf[0][0]))) return r;
      for (int t = 0; t < BP_NUM_TABLES; t++) {
        proof_digest(s->rec_cfg, chain_proof[t].data(), digest[t]);
        first_query_trace_path(s->rec_cfg, chain_proof[t].data(), leaf_cap[t], leaf_cap[t] + 4, &child_path[t][0], first_leaf[t]);
      }
    }
  }
  // root proof: the seven chains' digests, their seven (leaf digest, cap entry) pairs, the public values
  std::vector<uint64_t> pi;
  std::vector<PathWitness> root_paths(BP_NUM_TABLES);
  for (int t = 0; t < BP_NUM_TABLES; t++) pi.insert(pi.end(), digest[t], digest[t] + 4);
  for (int t = 0; t < BP_NUM_TABLES; t++) {
    pi.insert(pi.end(), leaf_cap[t], leaf_cap[t] + 8);
    root_paths[t] = std::move(child_path[t][0]);
  }
  pi.insert(pi.end(), pv.begin(), pv.end());
  if ((r = rec_prove(w, s->rec_cfg, s->special[0], pi, proof, &root_paths))) return r;
  if ((r = emit_box(0, CIRCUIT_ROOT, pi, proof, out, out_len))) return r;
  remember_proof(s, *out, *out_len);
  return BP_OK;
}
int bp_generate_txn_proof(const bp_state* s, const uint8_t* ir, size_t ir_len, const volatile int32_t* abort_flag,
                          uint8_t** out, size_t* out_len) try {
  return txn_proof_impl(s, ir, ir_len, abort_flag, nullptr, out, out_len);
}
BPG_ABI_CATCH("bp_generate_txn_proof")
// The same call with the reference's own flag type: Option<Arc<AtomicBool>> (proof_gen.rs:42) is one byte,
// `flag.as_ptr()` binds directly.
int bp_generate_txn_proof_u8(const bp_state* s, const uint8_t* ir, size_t ir_len, const volatile uint8_t* abort_flag,
                             uint8_t** out, size_t* out_len) try {
  return txn_proof_impl(s, ir, ir_len, nullptr, abort_flag, out, out_len);
}
BPG_ABI_CATCH("bp_generate_txn_proof_u8")
// generate_txn_proof where the transaction's Keccak table attests GIVEN hashing work: keccak_inputs = the states that
// go into its Keccak-f permutations (n_perms x 25 lanes, e.g. from bp_keccak256_permutation_inputs over the signed
// transaction and the contract code of its GenerationInputs).  abort_flag as in bp_generate_txn_proof_u8.
int bp_generate_txn_proof_keccak(const bp_state* s, const uint8_t* ir, size_t ir_len, const uint64_t* keccak_inputs,
                                 size_t n_perms, const volatile uint8_t* abort_flag, uint8_t** out, size_t* out_len) try {
  if (!keccak_inputs && n_perms) return fail(BP_ERR_INVALID_INPUT, "bp_generate_txn_proof_keccak: null inputs");
  static const uint64_t none = 0;
  TxnWitness wit;
  wit.in[3] = keccak_inputs ? keccak_inputs : &none;
  wit.n[3] = n_perms;
  return txn_proof_impl(s, ir, ir_len, nullptr, abort_flag, out, out_len, &wit);
}
BPG_ABI_CATCH("bp_generate_txn_proof_keccak")
// bp_txn_witness (include/bpg.h) -> the per-table form
static int witness_of(const bp_txn_witness* data, TxnWitness* wit) {
  static const uint64_t none = 0;
  const struct { int t; const uint64_t* p; size_t n; int given; } f[6] = {
      {4, data->sponge_rows, data->n_sponge_rows, data->has_keccak_sponge},
      {3, data->keccak_inputs, data->n_perms, data->has_keccak}, {5, data->logic_ops, data->n_logic_ops, data->has_logic},
      {6, data->memory_log, data->n_memory_ops, data->has_memory}, {0, data->arithmetic_ops, data->n_arithmetic_ops, data->has_arithmetic},
      {1, data->byte_sequences, data->n_byte_sequences, data->has_byte_packing}};
  for (const auto& x : f) {
    if (!x.given) continue;
    if (!x.p && x.n) return fail(BP_ERR_INVALID_INPUT, "bp_generate_txn_proof_witness: null data for table %s", TABLE_NAMES[x.t]);
    wit->in[x.t] = x.p ? x.p : &none;
    wit->n[x.t] = x.n;
  }
  return BP_OK;
}
// The general form: witness data for any of the tables that have an AIR (bp_txn_witness, include/bpg.h).
int bp_generate_txn_proof_witness(const bp_state* s, const uint8_t* ir, size_t ir_len, const bp_txn_witness* data,
                                  const volatile uint8_t* abort_flag, uint8_t** out, size_t* out_len) try {
  if (!data) return txn_proof_impl(s, ir, ir_len, nullptr, abort_flag, out, out_len);
  TxnWitness wit;
  int r = witness_of(data, &wit);
  if (r) return r;
  return txn_proof_impl(s, ir, ir_len, nullptr, abort_flag, out, out_len, &wit);
}
BPG_ABI_CATCH("bp_generate_txn_proof_witness")

// What upstream's `prove` yields before the recursion (AllProof): the seven table proofs of a transaction on their one
// transcript, with the public values and the lookup challenges.  data as for bp_generate_txn_proof_witness (nullable).
int bp_generate_txn_table_proofs(const bp_state* s, const uint8_t* ir, size_t ir_len, const bp_txn_witness* data,
                                 const volatile uint8_t* abort_flag, uint8_t** out, size_t* out_len) try {
  if (!data) return txn_proof_impl(s, ir, ir_len, nullptr, abort_flag, out, out_len, nullptr, true);
  TxnWitness wit;
  int r = witness_of(data, &wit);
  if (r) return r;
  return txn_proof_impl(s, ir, ir_len, nullptr, abort_flag, out, out_len, &wit, true);
}
BPG_ABI_CATCH("bp_generate_txn_table_proofs")

// verify_proof(all_stark, all_proof, config) of upstream, on the CPU: every table proof against the shared transcript
// (trace caps and public values observed, four lookup challenges drawn, then table after table), and the cross-table
// lookups between the tables that are proven with their AIRs (air::ctl).  cfg supplies the STARK parameters only.
// expect (nullable): the statement the caller wants proven -- per table the AIR, height and width, and the public values --
// as parse_ir derives it from the transaction's IR.  Without it the header of the blob is the PROVER's claim.
static int verify_table_proofs(const bp_config* cfg, const StarkCfg* expect, const uint64_t* expect_pv, const uint8_t* bytes, size_t len) {
  if (!cfg || !bytes) return fail(BP_ERR_INVALID_INPUT, "bp_verify_txn_table_proofs: null argument");
  if (len % 8 || len < (2 + BP_PV_WORDS + 4) * 8) return fail(BP_ERR_INVALID_INPUT, "table proofs: truncated");
  const uint64_t* W = reinterpret_cast<const uint64_t*>(bytes);
  const size_t n_words = len / 8;
  if (W[0] != TABLES_MAGIC || W[1] != BP_NUM_TABLES) return fail(BP_ERR_INVALID_INPUT, "table proofs: bad magic");
  const uint64_t* pv = W + 2;
  const uint64_t* ctl_in = pv + BP_PV_WORDS;
  for (size_t i = 0; i < BP_PV_WORDS + 4; i++) if (pv[i] >= gl::P) return fail(BP_ERR_VERIFY, "non-canonical public value or challenge");
  StarkCfg tcfg[BP_NUM_TABLES];
  std::vector<uint64_t> proof[BP_NUM_TABLES];
  size_t off = 2 + BP_PV_WORDS + 4;
  for (int t = 0; t < BP_NUM_TABLES; t++) {
    if (off + 4 > n_words) return fail(BP_ERR_INVALID_INPUT, "table proofs: truncated at table %s", TABLE_NAMES[t]);
    const uint64_t air_id = W[off], log_n = W[off + 1], n_cols = W[off + 2], pw = W[off + 3];
    off += 4;
    if (air_id >= air::COUNT || log_n > 30 || n_cols > 65536) return fail(BP_ERR_INVALID_INPUT, "table proofs: bad header of table %s", TABLE_NAMES[t]);
    if (air_id != air::SYNTHETIC && air_id != WITNESS_AIR[t] && !(t == 0 && air_id == air::ARITHMETIC_MUL))
      return fail(BP_ERR_VERIFY, "table %s is proven with AIR %llu, which is not that table's", TABLE_NAMES[t], (unsigned long long)air_id);
    tcfg[t] = table_cfg_of(*cfg, (uint32_t)log_n, (uint32_t)n_cols);
    tcfg[t].air_id = (uint32_t)air_id;
    if (expect && (expect[t].air_id != air_id || expect[t].log_n != log_n || expect[t].n_cols != n_cols))
      return fail(BP_ERR_VERIFY, "table %s is proven as AIR %llu, 2^%llu rows x %llu columns; the transaction's statement is AIR %u, 2^%u x %u "
                  "(a relabelled table would drop its constraints and its lookups)", TABLE_NAMES[t], (unsigned long long)air_id,
                  (unsigned long long)log_n, (unsigned long long)n_cols, expect[t].air_id, expect[t].log_n, expect[t].n_cols);
    int r = check_cfg(tcfg[t]);
    if (r) return r;
    if (pw != proof_layout(tcfg[t]).total || off + pw > n_words) return fail(BP_ERR_INVALID_INPUT, "table proofs: wrong length of table %s", TABLE_NAMES[t]);
    proof[t].assign(W + off, W + off + pw);
    off += pw;
  }
  if (off != n_words) return fail(BP_ERR_INVALID_INPUT, "table proofs: trailing words");
  if (expect_pv && std::memcmp(pv, expect_pv, BP_PV_WORDS * 8) != 0)
    return fail(BP_ERR_VERIFY, "the public values of the table proofs are not those of the transaction's IR");
  Challenger ch;
  for (int t = 0; t < BP_NUM_TABLES; t++) {
    const ProofLayout L = proof_layout(tcfg[t]);
    ch.observe(proof[t].data() + L.trace_cap, L.cap_words);
  }
  ch.observe(pv, BP_PV_WORDS);
  Ctl ctl;
  for (int i = 0; i < 4; i++) {
    ctl.v[i] = ch.challenge();
    if (ctl.v[i] != ctl_in[i]) return fail(BP_ERR_VERIFY, "the lookup challenges do not follow from the trace commitments");
  }
  for (int t = 0; t < BP_NUM_TABLES; t++) {
    int r = stark_verify(tcfg[t], nullptr, ctl, ch, proof[t].data(), proof[t].size());
    if (r) {
      const std::string why = bp_last_error();
      return fail(r, "table %s: %s", TABLE_NAMES[t], why.c_str());
    }
  }
  return check_lookups(tcfg, proof);
}
int bp_verify_txn_table_proofs(const bp_config* cfg, const uint8_t* bytes, size_t len) try {
  return verify_table_proofs(cfg, nullptr, nullptr, bytes, len);
}
BPG_ABI_CATCH("bp_verify_txn_table_proofs")
// verify_proof(all_stark, ...) where the VERIFIER fixes the statement, as upstream's does: which AIR proves each table,
// the table shapes and the public values come from the transaction's IR, not from the blob.
int bp_verify_txn_table_proofs_for(const bp_config* cfg, const uint8_t* ir, size_t ir_len, const uint8_t* bytes, size_t len) try {
  if (!cfg || !ir) return fail(BP_ERR_INVALID_INPUT, "bp_verify_txn_table_proofs_for: null argument");
  if (ir_len != BP_IR_WORDS * 8) return fail(BP_ERR_INVALID_INPUT, "IR must be %d bytes", BP_IR_WORDS * 8);
  StarkCfg expect[BP_NUM_TABLES];
  std::vector<uint64_t> pv;
  int r = parse_ir(*cfg, reinterpret_cast<const uint64_t*>(ir), nullptr, expect, &pv);
  if (r) return r;
  return verify_table_proofs(cfg, expect, pv.data(), bytes, len);
}
BPG_ABI_CATCH("bp_verify_txn_table_proofs_for")

int bp_generate_agg_proof(const bp_state* s, const uint8_t* lhs, size_t lhs_len, int lhs_is_agg,
                          const uint8_t* rhs, size_t rhs_len, int rhs_is_agg, uint8_t** out, size_t* out_len) try {
  if (!s || !lhs || !rhs || !out || !out_len) return fail(BP_ERR_INVALID_INPUT, "bp_generate_agg_proof: null argument");
  Box L, R;
  int r;
  if ((r = parse_box(lhs, lhs_len, s->rec_cfg, &L)) || (r = parse_box(rhs, rhs_len, s->rec_cfg, &R))) return r;
  if ((L.kind == 1) != (lhs_is_agg != 0) || (R.kind == 1) != (rhs_is_agg != 0) || L.kind > 1 || R.kind > 1)
    return fail(BP_ERR_INVALID_INPUT, "is_agg flags do not match the child proofs");
  // children must cover contiguous txn ranges (proof_types.rs:23-24)
  if (L.pv[1] != R.pv[0]) return fail(BP_ERR_INVALID_INPUT, "children are not contiguous: lhs ends at txn %llu, rhs starts at %llu",
                                      (unsigned long long)L.pv[1], (unsigned long long)R.pv[0]);
  if (L.pv[3] != R.pv[2] || std::memcmp(L.pv + 8, R.pv + 4, 32) != 0 || L.pv[12] != R.pv[12])
    return fail(BP_ERR_INVALID_INPUT, "children public values do not chain (gas / state root / block number)");
  {
    // A child this state produced is recognised by its hash and costs one Keccak; only two FOREIGN children (the
    // sub-block proofs gathered from other ranks, 10 ms of host Poseidon each) are verified side by side.  The
    // helper's message is thread-local, so it is carried over; the guard joins on every way out of the scope, so
    // an exception on this thread cannot reach a joinable std::thread's destructor (std::terminate).
    const bool l_known = produced_here(s, lhs, lhs_len), r_known = produced_here(s, rhs, rhs_len);
    int r_rhs = BP_OK;
    std::string lhs_err, rhs_err;
    std::thread helper;
    struct Joiner {
      std::thread& t;
      ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{helper};
    bool rhs_done = r_known;
    if (!l_known && !r_known) {
      try {
        helper = std::thread([&] {
          try {
            r_rhs = verify_foreign(s, R, "rhs child proof");
            if (r_rhs) rhs_err = bp_last_error();
          } catch (...) {
            r_rhs = BP_ERR_DEVICE;
            rhs_err = "rhs child proof: out of memory while verifying";
          }
        });
        rhs_done = true;
      } catch (const std::system_error&) {  // no thread to be had: verify in this one
      }
    }
    r = l_known ? BP_OK : verify_foreign(s, L, "lhs child proof");
    if (r) lhs_err = bp_last_error();
    if (helper.joinable()) helper.join();
    if (!rhs_done && (r_rhs = verify_foreign(s, R, "rhs child proof"))) rhs_err = bp_last_error();
    if (r) return fail(r, "%s", lhs_err.c_str());
    if (r_rhs) return fail(r_rhs, "%s", rhs_err.c_str());
  }
  std::vector<uint64_t> pi(AGG_PATH_PI0 + 16 + BP_PV_WORDS);
  proof_digest(s->rec_cfg, L.stark, &pi[0]);
  proof_digest(s->rec_cfg, R.stark, &pi[4]);
  pi[8] = lhs_is_agg != 0; pi[9] = rhs_is_agg != 0;
  // per child the opened trace row of its first query and the cap entry above it: the circuit walks the path between them
  std::vector<PathWitness> paths(2);
  first_query_trace_path(s->rec_cfg, L.stark, &pi[AGG_PATH_PI0], &pi[AGG_PATH_PI0 + 4], &paths[0]);
  first_query_trace_path(s->rec_cfg, R.stark, &pi[AGG_PATH_PI0 + 8], &pi[AGG_PATH_PI0 + 12], &paths[1]);
  uint64_t* pv = &pi[AGG_PATH_PI0 + 16];
  pv[0] = L.pv[0]; pv[1] = R.pv[1]; pv[2] = L.pv[2]; pv[3] = R.pv[3];
  std::memcpy(pv + 4, L.pv + 4, 32); std::memcpy(pv + 8, R.pv + 8, 32);
  pv[12] = L.pv[12];
  (void)hipSetDevice(s->cfg.device);
  WorkerLease lease(s);
  std::vector<uint64_t> proof;
  if ((r = rec_prove(*lease.w, s->rec_cfg, s->special[1], pi, proof, &paths))) return r;
  if ((r = emit_box(1, CIRCUIT_AGG, pi, proof, out, out_len))) return r;
  remember_proof(s, *out, *out_len);
  return BP_OK;
}
BPG_ABI_CATCH("bp_generate_agg_proof")

int bp_generate_block_proof(const bp_state* s, const uint8_t* parent, size_t parent_len, const uint8_t* agg,
                            size_t agg_len, uint8_t** out, size_t* out_len, uint64_t* b_height) try {
  if (!s || !agg || !out || !out_len) return fail(BP_ERR_INVALID_INPUT, "bp_generate_block_proof: null argument");
  Box A, Pb;
  int r;
  if ((r = parse_box(agg, agg_len, s->rec_cfg, &A))) return r;
  if (A.kind != 1) return fail(BP_ERR_INVALID_INPUT, "curr_block_agg_proof is not an aggregation proof");
  std::vector<uint64_t> pi(BLOCK_PATH_PI0 + 8 + BP_PV_WORDS, 0);
  if (parent) {
    if ((r = parse_box(parent, parent_len, s->rec_cfg, &Pb))) return r;
    if (Pb.kind != 2) return fail(BP_ERR_INVALID_INPUT, "parent is not a block proof");
    if (Pb.pv[12] + 1 != A.pv[12]) return fail(BP_ERR_INVALID_INPUT, "parent block height %llu does not precede %llu",
                                                (unsigned long long)Pb.pv[12], (unsigned long long)A.pv[12]);
    if ((r = verify_child(s, Pb, "parent block proof", parent, parent_len))) return r;
    proof_digest(s->rec_cfg, Pb.stark, &pi[0]);
    pi[8] = 1;
  }
  if ((r = verify_child(s, A, "curr_block_agg_proof", agg, agg_len))) return r;
  proof_digest(s->rec_cfg, A.stark, &pi[4]);
  std::vector<PathWitness> paths(1);
  first_query_trace_path(s->rec_cfg, A.stark, &pi[BLOCK_PATH_PI0], &pi[BLOCK_PATH_PI0 + 4], &paths[0]);
  std::memcpy(&pi[BLOCK_PATH_PI0 + 8], A.pv, BP_PV_WORDS * 8);
  (void)hipSetDevice(s->cfg.device);
  WorkerLease lease(s);
  std::vector<uint64_t> proof;
  if ((r = rec_prove(*lease.w, s->rec_cfg, s->special[2], pi, proof, &paths))) return r;
  if (b_height) *b_height = A.pv[12];  // block_metadata.block_number.low_u64(), proof_gen.rs:90-94
  if ((r = emit_box(2, CIRCUIT_BLOCK, pi, proof, out, out_len))) return r;
  remember_proof(s, *out, *out_len);
  return BP_OK;
}
BPG_ABI_CATCH("bp_generate_block_proof")

int bp_verifier_state_from_prover(const bp_state* s, bp_verifier_state** out) try {
  if (!s || !out) return fail(BP_ERR_INVALID_INPUT, "bp_verifier_state_from_prover: null argument");
  bp_verifier_state* v = new bp_verifier_state();
  v->rec_cfg = s->rec_cfg;
  for (int k = 0; k < 3; k++) {
    v->special[k].cap = s->special[k].consts.cap;
    std::memcpy(v->special[k].digest, s->special[k].digest, 32);
  }
  *out = v;
  return BP_OK;
}
BPG_ABI_CATCH("bp_verifier_state_from_prover")
// ProverStateBuilder::build_verifier (verifier_state.rs:34-42): build the circuits, keep the light part.
int bp_verifier_state_build(const bp_config* cfg, bp_verifier_state** out) try {
  bp_state* s = nullptr;
  bp_config c = *cfg;
  c.n_workers = 1;
  int r = bp_state_build(&c, &s);
  if (r) return r;
  r = bp_verifier_state_from_prover(s, out);
  bp_state_free(s);
  return r;
}
BPG_ABI_CATCH("bp_verifier_state_build")
// Light verifier data from raw caps (what a verifier-only deployment would load from disk):
// caps = root, agg, block constants caps, each 4 << stark_cap_height words.  CPU only.
int bp_verifier_state_from_caps(const bp_config* cfg, const uint64_t* caps, bp_verifier_state** out) try {
  if (!cfg || !caps || !out) return fail(BP_ERR_INVALID_INPUT, "bp_verifier_state_from_caps: null argument");
  const StarkCfg rc = rec_cfg_of(*cfg);
  int r = check_cfg(rc);
  if (r) return r;
  const size_t cw = (size_t)4 << rc.cap_height;
  for (size_t i = 0; i < 3 * cw; i++) if (caps[i] >= gl::P) return fail(BP_ERR_INVALID_INPUT, "non-canonical cap word");
  bp_verifier_state* v = new bp_verifier_state();
  v->rec_cfg = rc;
  for (int k = 0; k < 3; k++) {
    v->special[k].cap.assign(caps + k * cw, caps + (k + 1) * cw);
    hash_no_pad_host(v->special[k].cap.data(), cw, v->special[k].digest);
  }
  *out = v;
  return BP_OK;
}
BPG_ABI_CATCH("bp_verifier_state_from_caps")
void bp_verifier_state_free(bp_verifier_state* v) { delete v; }

int bp_verify_proof(const bp_verifier_state* v, const uint8_t* proof, size_t len) try {
  if (!v || !proof) return fail(BP_ERR_INVALID_INPUT, "bp_verify_proof: null argument");
  Box b;
  int r = parse_box(proof, len, v->rec_cfg, &b);
  if (r) return r;
  if (b.circuit != CIRCUIT_ROOT + b.kind) return fail(BP_ERR_VERIFY, "proof was made by circuit %llu, expected %u",
                                                      (unsigned long long)b.circuit, CIRCUIT_ROOT + (uint32_t)b.kind);
  for (size_t i = 0; i < b.n_pi; i++) if (b.pi[i] >= gl::P) return fail(BP_ERR_VERIFY, "non-canonical public input");
  return rec_verify(v->rec_cfg, v->special[b.kind], b);
}
BPG_ABI_CATCH("bp_verify_proof")
int bp_verify_block_proof(const bp_verifier_state* v, const uint8_t* proof, size_t len) try {
  if (!v || !proof) return fail(BP_ERR_INVALID_INPUT, "bp_verify_block_proof: null argument");
  Box b;
  int r = parse_box(proof, len, v->rec_cfg, &b);
  if (r) return r;
  if (b.kind != 2) return fail(BP_ERR_VERIFY, "not a block proof");
  return bp_verify_proof(v, proof, len);
}
BPG_ABI_CATCH("bp_verify_block_proof")

}  // extern "C"
