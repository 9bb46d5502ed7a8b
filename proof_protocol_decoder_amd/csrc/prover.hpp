// prover.hpp -- host side of the per-table STARK prover: transcript, device arena, commitments.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstring>
#include <vector>
#include "common.hpp"
#include "gl.hpp"
#include "stark_kernels.hpp"

namespace bpg {

// ---- host Poseidon (transcript only: K7 stays on the host) ----
void poseidon_host(uint64_t s[12]);
void hash_no_pad_host(const uint64_t* in, size_t len, uint64_t out[4]);
// the same hash with the witness of its in-circuit computation: ceil(n / 8) rows of air::plonk::H_WIRES wires (AIR 8)
void poseidon_hash_rows(const uint64_t* in, size_t n, std::vector<uint64_t>* rows, uint64_t digest[4]);
void poseidon_merkle_rows(const uint64_t leaf[4], uint64_t index, const uint64_t* siblings, uint32_t depth, uint64_t* rows,
                          uint64_t root[4]);

// plonky2::iop::challenger::Challenger: overwrite-mode duplex sponge, outputs popped from the back.
class Challenger {
 public:
  Challenger() { std::memset(this, 0, sizeof(*this)); }
  void observe(uint64_t e) {
    n_out_ = 0;
    in_[n_in_++] = e;
    if (n_in_ == 8) duplex();
  }
  void observe(const uint64_t* e, size_t n) {
    for (size_t i = 0; i < n; i++) observe(e[i]);
  }
  void observe(gl::Ext e) {
    observe(e.c0);
    observe(e.c1);
  }
  uint64_t challenge() {
    if (n_in_ || !n_out_) duplex();
    return out_[--n_out_];
  }
  gl::Ext challenge_ext() {
    uint64_t a = challenge(), b = challenge();
    return gl::Ext{a, b};
  }
  // state a PoW candidate is spliced into (fri_proof_of_work's duplex_intermediate_state)
  void pow_state(uint64_t st[12], uint32_t* pos) const {
    std::memcpy(st, state_, sizeof(state_));
    for (unsigned i = 0; i < n_in_; i++) st[i] = in_[i];
    *pos = n_in_;
  }

 private:
  void duplex() {
    for (unsigned i = 0; i < n_in_; i++) state_[i] = in_[i];
    n_in_ = 0;
    poseidon_host(state_);
    std::memcpy(out_, state_, 8 * sizeof(uint64_t));
    n_out_ = 8;
  }
  uint64_t state_[12], in_[8], out_[8];
  unsigned n_in_, n_out_;
};

// ---- shapes ----
struct StarkCfg {
  uint32_t log_n, n_cols, n_const, deg_pow, rate_bits, cap_height, num_queries, pow_bits, arity_bits,
      final_poly_bits;
  uint32_t air_id = air::SYNTHETIC;  // which AIR the table proves (air.hpp); header word 14 of the proof
};
struct ProofLayout {
  size_t cap_words, trace_cap, aux_cap, quot_cap, open_zeta, open_next, open_first, fri_caps, final_poly, pow,
      queries, query_words, total;
  uint32_t n_aux, n_quot, n_layers, final_len, n_zeta, n_next, depth0;
};
constexpr uint64_t PROOF_MAGIC = 0x4B52415453475042ULL;  // "BPGSTARK"
constexpr size_t PROOF_HDR_WORDS = 16;
extern std::atomic<int> g_k5_spread_all;
int check_cfg(const StarkCfg& c);
ProofLayout proof_layout(const StarkCfg& c);
void proof_digest(const StarkCfg& c, const uint64_t* proof, uint64_t out[4]);

// ---- device memory: one bump arena per worker, no hipMalloc/hipFree on the proving path ----
class DeviceArena {
 public:
  int init(size_t bytes);
  void destroy();
  // returns nullptr when exhausted (caller reports BP_ERR_DEVICE)
  uint64_t* alloc_words(size_t words);
  size_t mark() const { return off_; }
  void release(size_t m) { off_ = m; }
  size_t capacity() const { return cap_; }
  size_t high_water() const { return high_; }

 private:
  char* base_ = nullptr;
  size_t cap_ = 0, off_ = 0, high_ = 0;
};

constexpr size_t HASH_ROWS_WORDS = (size_t)(air::plonk::HASH_ROWS_MAX + air::plonk::MERKLE_ROWS_MAX + air::plonk::LEAF_ROWS_MAX) * air::plonk::H_WIRES;  // per proof: list rows, Merkle rows, leaf rows

struct Committed {
  uint64_t *coeffs = nullptr, *lde = nullptr, *digests = nullptr;
  const uint64_t* values = nullptr;  // the committed values on the trace domain, while the caller keeps them (else null)
  uint32_t log_n = 0, n_cols = 0, rate_bits = 0, cap_height = 0;
  std::vector<uint64_t> cap;  // host copy, 2^cap_height * 4 words
};

// One per host thread that proves: its own stream, arena and pinned staging buffer.
struct Worker {
  hipStream_t stream = nullptr;
  DeviceArena arena;
  uint64_t* pinned = nullptr;      // host staging / mailbox (kernels may write it directly)
  uint64_t* pinned_dev = nullptr;  // the same memory as seen from the device
  size_t pinned_words = 0;
  // the hash-row witnesses of a batch of recursion-shaped proofs (AIR 8), made on the host, read by the witness kernel:
  // pinned memory of its own (the mailbox above belongs to the kernels that mirror caps and openings into it)
  uint64_t *hash_rows = nullptr, *hash_rows_dev = nullptr;
  unsigned long long* d_pow_result = nullptr;  // MAX_BATCH words: one witness per proof of a batch
  hipEvent_t sync_event = nullptr;  // blocking-sync event: waiting threads sleep instead of spinning
  const volatile int32_t* abort_flag = nullptr;
  const volatile uint8_t* abort_flag_u8 = nullptr;  // AtomicBool::as_ptr() of the reference's Arc<AtomicBool>
  int device = 0;

  int init(int device, size_t arena_bytes);
  void destroy();
  int d2h(uint64_t* host_dst, const uint64_t* dev_src, size_t words);  // async copy + stream sync
  int wait();  // everything queued on the stream has completed; the thread sleeps meanwhile (prover.cpp, "Host waits")
  int wait_recorded();  // the same for a sync_event the caller has recorded already
  bool aborted() const { return (abort_flag && *abort_flag) || (abort_flag_u8 && *abort_flag_u8); }
};

// PolynomialBatch::from_values / from_coeffs: LDE + Merkle.  d_in is n_cols x n (values natural, or
// bit-reversed coefficients when from_coeffs; then the commitment aliases d_in as its coefficients).
int commit(Worker& w, const uint64_t* d_in, uint32_t n_cols, uint32_t log_n, uint32_t rate_bits,
           uint32_t cap_height, bool from_coeffs, Committed* out);
// `batch` (<= MAX_BATCH) commitments of one shape, every launch covering all of them: d_in = [batch][n_cols][n]
int commit_batch(Worker& w, const uint64_t* d_in, uint32_t n_cols, uint32_t batch, uint32_t log_n, uint32_t rate_bits,
                 uint32_t cap_height, bool from_coeffs, Committed* out);

// The two halves of commit_batch: the launches (arena of `mem`, stream and cap mailbox of `lane`, caps from word
// `mailbox_word` of the mailbox on) and the wait for the caps.  Commitments that do not depend on each other can be
// launched on different lanes and finished afterwards (proofgen.cpp: the trace commitments of a lone transaction).
struct PendingCommit {
  Worker* lane = nullptr;
  size_t mailbox_word = 0;
  const uint64_t* d_in = nullptr;
  uint64_t *coeffs = nullptr, *lde = nullptr, *digests = nullptr;
  size_t dw = 0, cw = 0;
  uint32_t n_cols = 0, batch = 0, log_n = 0, rate_bits = 0, cap_height = 0;
  bool from_coeffs = false;
};
int commit_launch(Worker& mem, Worker& lane, size_t mailbox_word, const uint64_t* d_in, uint32_t n_cols, uint32_t batch,
                  uint32_t log_n, uint32_t rate_bits, uint32_t cap_height, bool from_coeffs, PendingCommit* pc);
int commit_finish(const PendingCommit& pc, Committed* out);

// prove_single_table on the synthetic AIR.  The caller has already observed the trace cap(s) and
// drawn ctl (plonky2_evm prover order).  Fills `proof` (proof_layout(cfg).total words).
// first_trace_leaf (nullable, 4 words per proof): the digest of the trace-oracle leaf the proof's first query opens -- what
// a parent recursion circuit's Merkle path starts from; gathered on the device with the query openings.
int stark_prove(Worker& w, const StarkCfg& cfg, const Committed* consts, const Committed& trace,
                const uint64_t* d_trace_values, const Ctl& ctl, Challenger& ch, std::vector<uint64_t>& proof,
                uint64_t* first_trace_leaf = nullptr);
// The same for `batch` (<= MAX_BATCH, batch * num_queries <= MAX_BATCH_QUERIES) proofs of ONE shape in lock-step:
// independent transcripts, every kernel launch and every host wait shared (the seven per-table recursion chains of
// a transaction, proofgen.cpp).  Proof b's bytes are those stark_prove would give for (consts[b], trace[b], ...).
// consts: per proof, may be null when the shape has no constant columns.
int stark_prove_batch(Worker& w, const StarkCfg& cfg, uint32_t batch, const Committed* const* consts,
                      const Committed* trace, const uint64_t* const* d_trace_values, const Ctl* ctl, Challenger* ch,
                      std::vector<uint64_t>* proofs, uint64_t* first_trace_leaf = nullptr);

void tune_host_wait(int mode);  // bp_tune_host_wait
void tune_host_poseidon(int mode);  // bp_tune_host_poseidon
// hash_kernels.hip: +1 / -1 as a prover starts / finishes (the Poseidon kernel choice follows the load)
void prover_active(int delta);
int provers_active();  // how many are at work right now
bool device_loaded();  // six or more provers at work on the device

// launch-argument builders shared by the prover and the L0 entry points (stark_api.cpp)
// fills everything but the matrix pointers, apow (2 * n_constraints words) and partial (quotient_partial_words).
// loaded: how the units are spread over workgroup rows -- -1 by the device's load right now, 0 / 1 stated (a caller
// that sizes a buffer in one call and launches in another must not depend on a state that can flip in between)
// coset (nullable): the per-coset constants launch_quotient needs; batch: proofs of this shape in the launch
int quotient_args(const StarkCfg& cfg, const Ctl& ctl, uint64_t alpha0, uint64_t alpha1, QuotArgs* out, QuotCoset* coset,
                  int loaded = -1, uint32_t batch = 1);
size_t quotient_partial_words(const QuotArgs& qa);  // 0 when the table takes one pass
int fri_layer_args(uint32_t log_nl, uint32_t rate_bits, uint32_t arity_bits, uint64_t shift, FriLayerArgs* out);

// CPU verifier (verifier.cpp).  The caller has driven `ch` through the same prologue as the prover.
int stark_verify(const StarkCfg& cfg, const uint64_t* const_cap, const Ctl& ctl, Challenger& ch,
                 const uint64_t* proof, size_t n_words);

}  // namespace bpg
