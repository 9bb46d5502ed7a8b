/* oracle/proofgen.c -- txn / agg / block proof composition (CPU restatement of the synthetic
 * workload behind plonky_block_proof_gen/src/proof_gen.rs:39-110).
 * TEST INFRASTRUCTURE ONLY; "parity unpinned" by the reference (see gl.h).
 * Produces byte-identical containers to libbpg's bp_generate_{txn,agg,block}_proof.
 */
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NUM_TABLES 7
#define IR_WORDS 25
#define PV_WORDS 13
#define BOX_HDR 4
#define IR_MAGIC 0x52494E5854475042ULL
#define BOX_MAGIC 0x464F4F5250475042ULL
#define CIRCUIT_ROOT 7

typedef struct {
  uint32_t table_log_lo[NUM_TABLES], table_log_hi[NUM_TABLES];
  uint32_t stark_rate_bits, stark_cap_height, stark_num_queries, stark_pow_bits, arity_bits, final_poly_bits;
  uint32_t rec_log_n, rec_n_cols, rec_n_const, rec_rate_bits, rec_num_queries, rec_pow_bits;
  uint32_t shrink_depth;
  uint32_t rec_air_id; /* 0: recursion-shaped proofs on the synthetic AIR; 8: on the PLONK-shaped circuit (plonk_air.c) */
} orc_pg_config;

typedef struct {
  int built;
  gl_t* const_values;
  orc_committed* consts;
  gl_t digest[4];
  unsigned n_paths, path_pi0, depth, leaf_len; /* the Merkle paths the circuit walks, the leaves it hashes (plonk_air.c) */
} circuit_t;

typedef struct orc_pg_state {
  orc_pg_config cfg;
  orc_stark_cfg rec;
  circuit_t table[NUM_TABLES][32];
  circuit_t shrink[NUM_TABLES];
  circuit_t special[3];
} orc_pg_state;

static uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static uint64_t circuit_seed(uint32_t kind, uint32_t degree) {
  return splitmix64(0xC12C0175EEDULL + ((uint64_t)kind << 8) + degree);
}
static orc_stark_cfg rec_cfg_of(const orc_pg_config* c) {
  orc_stark_cfg r = {c->rec_log_n, c->rec_n_cols, c->rec_n_const, 3, c->rec_rate_bits, c->stark_cap_height,
                     c->rec_num_queries, c->rec_pow_bits, c->arity_bits, c->final_poly_bits, c->rec_air_id, {0, 0, 0, 0}};
  return r;
}
static orc_stark_cfg table_cfg_of(const orc_pg_config* c, uint32_t log_n, uint32_t width) {
  orc_stark_cfg r = {log_n, width, 0, 1, c->stark_rate_bits, c->stark_cap_height, c->stark_num_queries,
                     c->stark_pow_bits, c->arity_bits, c->final_poly_bits, ORC_AIR_SYNTHETIC, {0, 0, 0, 0}};
  return r;
}

orc_pg_state* orc_pg_state_build(const orc_pg_config* cfg) {
  orc_pg_state* s = (orc_pg_state*)calloc(1, sizeof(*s));
  s->cfg = *cfg;
  s->rec = rec_cfg_of(cfg);
  return s;
}
static void circuit_free(circuit_t* c) {
  if (c->built) { free(c->const_values); orc_committed_free(c->consts); }
}
void orc_pg_state_free(orc_pg_state* s) {
  if (!s) return;
  for (int t = 0; t < NUM_TABLES; t++) for (int d = 0; d < 32; d++) circuit_free(&s->table[t][d]);
  for (int k = 0; k < 3; k++) circuit_free(&s->special[k]);
  for (int t = 0; t < NUM_TABLES; t++) circuit_free(&s->shrink[t]);
  free(s);
}
/* circuits are preprocessed lazily (same data as libbpg's eager bp_state_build) */
static circuit_t* get_circuit_leaf(orc_pg_state* s, circuit_t* c, uint64_t seed, unsigned pi_len, unsigned n_paths, unsigned depth, unsigned path_pi0, unsigned leaf_len);
static circuit_t* get_circuit_depth(orc_pg_state* s, circuit_t* c, uint64_t seed, unsigned pi_len, unsigned n_paths, unsigned depth, unsigned path_pi0) {
  return get_circuit_leaf(s, c, seed, pi_len, n_paths, depth, path_pi0, 0);
}
static circuit_t* get_circuit_leaf(orc_pg_state* s, circuit_t* c, uint64_t seed, unsigned pi_len, unsigned n_paths, unsigned depth, unsigned path_pi0, unsigned leaf_len) {
  if (!c->built) {
    size_t n = (size_t)1 << s->rec.log_n;
    c->const_values = (gl_t*)malloc(s->rec.n_const * n * sizeof(gl_t));
    c->n_paths = n_paths; c->path_pi0 = path_pi0; c->depth = depth; c->leaf_len = leaf_len;
    if (s->rec.air_id == ORC_AIR_PLONK)
      orc_plonk_constants(seed, s->rec.log_n, pi_len, n_paths, depth, path_pi0, leaf_len, c->const_values);
    else orc_synth_constants(seed, s->rec.log_n, s->rec.n_const, c->const_values);
    c->consts = orc_commit_values(c->const_values, s->rec.log_n, s->rec.n_const, s->rec.rate_bits, s->rec.cap_height);
    orc_hash_no_pad(orc_committed_cap(c->consts), (size_t)4 << s->rec.cap_height, c->digest);
    c->built = 1;
  }
  return c;
}
/* the length of the public-input list each circuit hashes in-circuit: a table's chain circuits (digest, table, depth) 6;
 * root 7 digests + the public values; aggregation two digests, two flags + the public values; block two digests, a flag + them */
/* Every recursion circuit walks one Merkle path per child (the child's first trace opening: leaf digest and cap entry are
 * eight more words of the list).  A table's chain: level 0 by the circuit of the table's height (its child is the table's
 * STARK proof: d + rate - cap levels), the levels above by the table's shrink circuit (a recursion-shaped child). */
enum { CHAIN_PATH_AT = 6, CHAIN_PI = 6 + 8, ROOT_PATHS_AT = 4 * NUM_TABLES, ROOT_PI = 4 * NUM_TABLES + 8 * NUM_TABLES + PV_WORDS };
static circuit_t* get_circuit_depth(orc_pg_state* s, circuit_t* c, uint64_t seed, unsigned pi_len, unsigned n_paths, unsigned depth, unsigned path_pi0);
static circuit_t* get_circuit_leaf(orc_pg_state* s, circuit_t* c, uint64_t seed, unsigned pi_len, unsigned n_paths, unsigned depth, unsigned path_pi0, unsigned leaf_len);
static circuit_t* table_circuit(orc_pg_state* s, int t, uint32_t d) {
  return get_circuit_depth(s, &s->table[t][d], circuit_seed(t, d), CHAIN_PI, 1, d + s->cfg.stark_rate_bits - s->cfg.stark_cap_height, CHAIN_PATH_AT);
}
/* A circuit whose children are recursion-shaped proofs also hashes the row each child opens, where it has the rows for it:
 * 17 list rows at most, the Merkle rows, ceil(n_cols / 8) leaf rows per path (120 at most), and one arithmetic group of four */
static unsigned leaf_len_of(const orc_pg_state* s, unsigned n_paths, unsigned depth) {
  if (s->rec.air_id != ORC_AIR_PLONK || s->rec.n_cols <= 8) return 0;
  const unsigned lh = (s->rec.n_cols + 7) / 8, rows = 17 + n_paths * depth + n_paths * lh;
  return (n_paths * lh <= 120 && (rows + 3) / 4 * 4 + 4 <= (1u << s->rec.log_n)) ? s->rec.n_cols : 0;
}
static circuit_t* shrink_circuit(orc_pg_state* s, int t) {
  const unsigned depth = s->rec.log_n + s->rec.rate_bits - s->rec.cap_height;
  return get_circuit_leaf(s, &s->shrink[t], circuit_seed(t, 255), CHAIN_PI, 1, depth, CHAIN_PATH_AT, leaf_len_of(s, 1, depth));
}
/* The aggregation circuit also walks, per child, the Merkle path of the child's first trace opening (its leaf digest and
 * the cap entry above it: eight more words of the list per child, after the digests and flags); the block circuit walks
 * its aggregation child's. */
enum { AGG_PATHS_AT = 10, BLOCK_PATH_AT = 9, AGG_PI = 10 + 16 + PV_WORDS, BLOCK_PI = 9 + 8 + PV_WORDS };
static circuit_t* special_circuit(orc_pg_state* s, int k) {
  static const unsigned PI_LEN[3] = {ROOT_PI, AGG_PI, BLOCK_PI}, PATHS[3] = {NUM_TABLES, 2, 1}, AT[3] = {ROOT_PATHS_AT, AGG_PATHS_AT, BLOCK_PATH_AT};
  const unsigned depth = s->rec.log_n + s->rec.rate_bits - s->rec.cap_height;
  const unsigned leaf_len = leaf_len_of(s, PATHS[k], depth);
  return get_circuit_leaf(s, &s->special[k], circuit_seed(CIRCUIT_ROOT + k, 0), PI_LEN[k], PATHS[k], depth, AT[k], leaf_len);
}

/* paths: the witness of the circuit's Merkle paths (circ->n_paths of them, 1 + 4 depth words each), or NULL */
static int rec_prove(orc_pg_state* s, circuit_t* circ, const gl_t* pi, size_t n_pi, const gl_t* paths, gl_t* proof) {
  gl_t pi_hash[4];
  orc_hash_no_pad(pi, n_pi, pi_hash);
  orc_challenger ch;
  orc_ch_init(&ch);
  orc_ch_observe_many(&ch, circ->digest, 4);
  orc_ch_observe_many(&ch, pi_hash, 4);
  size_t n = (size_t)1 << s->rec.log_n;
  gl_t* trace = (gl_t*)malloc(s->rec.n_cols * n * sizeof(gl_t));
  orc_stark_cfg rcfg = s->rec; /* the PLONK-shaped circuit binds the hash of the public inputs to its first row */
  if (rcfg.air_id == ORC_AIR_PLONK) {
    memcpy(rcfg.pub, pi_hash, sizeof(rcfg.pub));
    orc_plonk_trace(pi_hash[0], pi, (unsigned)n_pi, circ->n_paths, circ->depth, circ->path_pi0, circ->leaf_len, paths, circ->const_values, rcfg.log_n, trace);
  } else {
    orc_synth_trace(pi_hash[0], &s->rec, circ->const_values, trace);
  }
  orc_committed* tc = orc_commit_values(trace, s->rec.log_n, s->rec.n_cols, s->rec.rate_bits, s->rec.cap_height);
  orc_ch_observe_many(&ch, orc_committed_cap(tc), (size_t)4 << s->rec.cap_height);
  gl_t ctl[4];
  for (int i = 0; i < 4; i++) ctl[i] = orc_ch_challenge(&ch);
  int rc = orc_stark_prove(&rcfg, circ->consts, tc, trace, ctl, &ch, proof);
  orc_committed_free(tc);
  free(trace);
  return rc;
}

static gl_t* emit_box(uint64_t kind, const gl_t* pi, size_t n_pi, const gl_t* stark, size_t sw, size_t* out_words) {
  gl_t* o = (gl_t*)malloc((BOX_HDR + n_pi + sw) * sizeof(gl_t));
  o[0] = BOX_MAGIC; o[1] = kind; o[2] = n_pi; o[3] = CIRCUIT_ROOT + kind;
  memcpy(o + BOX_HDR, pi, n_pi * 8);
  memcpy(o + BOX_HDR + n_pi, stark, sw * 8);
  *out_words = BOX_HDR + n_pi + sw;
  return o;
}

/* Preprocess now what a txn proof of this IR will touch (its seven table circuits and the root circuit), as
 * libbpg's eager bp_state_build has done before any proof is timed.  Lets a caller time orc_pg_txn alone. */
int orc_pg_preprocess(orc_pg_state* s, const uint64_t* I) {
  if (I[0] != IR_MAGIC) return -2;
  for (int t = 0; t < NUM_TABLES; t++) {
    if (I[11 + t] < s->cfg.table_log_lo[t] || I[11 + t] >= s->cfg.table_log_hi[t]) return -3;
    table_circuit(s, t, (uint32_t)I[11 + t]);
  }
  special_circuit(s, 0);
  return 0;
}

/* generate_txn_proof (proof_gen.rs:39-56) on the synthetic workload */
/* witness data per table (NULL: drawn from the seed): items[t] x n[t], WIT_WORDS[t] words each */
typedef struct { const uint64_t* items[NUM_TABLES]; size_t n[NUM_TABLES]; } pg_witness;
static const unsigned WIT_WORDS[NUM_TABLES] = {9, 6, 0, 25, 44, 9, 11};
int orc_pg_check_lookups(const orc_stark_cfg tcfg[NUM_TABLES], const gl_t* const proofs[NUM_TABLES]);
static int pg_txn(orc_pg_state* s, const uint64_t* I, const pg_witness* wit, gl_t** out, size_t* out_words);
int orc_pg_txn(orc_pg_state* s, const uint64_t* I, gl_t** out, size_t* out_words) { return pg_txn(s, I, NULL, out, out_words); }
/* the same with the Keccak table's permutation inputs given (n_perms x 25 lanes; the rest of the table: zero states) */
int orc_pg_txn_keccak(orc_pg_state* s, const uint64_t* I, const uint64_t* keccak_inputs, size_t n_perms, gl_t** out,
                      size_t* out_words) {
  static const uint64_t none = 0;
  pg_witness w;
  memset(&w, 0, sizeof(w));
  w.items[3] = keccak_inputs ? keccak_inputs : &none;
  w.n[3] = n_perms;
  return pg_txn(s, I, &w, out, out_words);
}
/* the general form: data for the tables with an AIR, by table index (0 arithmetic [n][9], 1 byte packing [n][6],
 * 3 Keccak [n][25], 5 logic [n][9], 6 memory [n][11] sorted); given[t] != 0 selects a table (n may be 0) */
int orc_pg_txn_witness(orc_pg_state* s, const uint64_t* I, const uint64_t* const items[NUM_TABLES], const size_t n[NUM_TABLES],
                       const int given[NUM_TABLES], gl_t** out, size_t* out_words) {
  static const uint64_t none = 0;
  pg_witness w;
  memset(&w, 0, sizeof(w));
  for (int t = 0; t < NUM_TABLES; t++)
    if (given[t] && WIT_WORDS[t]) {
      w.items[t] = items[t] ? items[t] : &none;
      w.n[t] = n[t];
    }
  return pg_txn(s, I, &w, out, out_words);
}
/* the table's whole input list: the given items, then padding (zero states / rows without an operation; the memory
 * log goes on reading its last cell, one tick later each row) */
static uint64_t* padded_items(int t, size_t rows, const uint64_t* items, size_t n) {
  const size_t cap = t == 3 ? (rows + 23) / 24 : rows, wds = WIT_WORDS[t];
  uint64_t* in = (uint64_t*)calloc(cap * wds, 8);
  memcpy(in, items, n * wds * 8);
  if (t == 6) {
    uint64_t last[11] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (n) memcpy(last, items + (n - 1) * 11, sizeof(last));
    last[0] = 1;
    for (size_t i = n; i < cap; i++) {
      last[2]++;
      memcpy(in + i * 11, last, sizeof(last));
    }
  }
  return in;
}
/* Test hook: 0 makes the PROVER skip its own lookup check, so that tests can hand table proofs that do not form one
 * statement to the verifiers (which always check). */
static int g_prover_checks_lookups = 1;
void orc_pg_set_prover_lookup_check(int on) { g_prover_checks_lookups = on != 0; }

/* The seven table proofs of a transaction on their one transcript (upstream's `prove`: AllProof), their shapes, the
 * public values and the lookup challenges.  proofs[t] is malloc'ed. */
static int pg_tables(orc_pg_state* s, const uint64_t* I, const pg_witness* wit, orc_stark_cfg tcfg[NUM_TABLES],
                     gl_t pv[PV_WORDS], gl_t ctl[4], gl_t* proofs[NUM_TABLES]) {
  const orc_pg_config* cfg = &s->cfg;
  /* version 1: a transaction; version 2: a dummy entry (decoding.rs:484-520: "Txn numbers before/after",
   * "Gas used before/after" equal, tries unchanged) -- same tables proven, public values do not advance */
  /* flags above the version byte: 0x100 = the Keccak table (index 3, prover_state.rs:85-93) is proven with the
   * Keccak-f AIR (keccak_air.c) instead of the synthetic one: 2430 columns, witness drawn from the seed;
   * 0x200 = the logic table (index 5) is proven with the logic AIR (logic_air.c): 523 columns;
   * 0x400 = the memory table (index 6) with the memory AIR (memory_air.c): 44 columns;
   * 0x800 = the arithmetic table (index 0) with the arithmetic AIR (arithmetic_air.c): 309 columns;
   * 0x1000 = the byte-packing table (index 1) with the byte-packing AIR (byte_packing_air.c): 299 columns;
   * 0x2000 = the Keccak sponge table (index 4) with the Keccak sponge AIR (keccak_sponge_air.c): 2414 columns */
  const uint64_t ver = I[1] & 0xFF, flags = I[1] >> 8;
  if (I[0] != IR_MAGIC || (ver != 1 && ver != 2) || flags > 127 || ((flags & 8) && (flags & 64))) return -2;
  const int mul_air = (int)((flags >> 6) & 1); /* 0x4000: the arithmetic table by the multiplication AIR (arithmetic_mul_air.c): 1217 columns */
  const int dummy = ver == 2, keccak_air = (int)(flags & 1), logic_air = (int)((flags >> 1) & 1), memory_air = (int)((flags >> 2) & 1),
            arithmetic_air = (int)((flags >> 3) & 1), byte_packing_air = (int)((flags >> 4) & 1),
            sponge_air = (int)((flags >> 5) & 1);
  if (dummy && I[4] != I[5]) return -2;
  for (int t = 0; t < NUM_TABLES; t++) {
    if (I[11 + t] < cfg->table_log_lo[t] || I[11 + t] >= cfg->table_log_hi[t]) return -3;
    tcfg[t] = table_cfg_of(cfg, (uint32_t)I[11 + t], (uint32_t)I[18 + t]);
  }
  if (keccak_air) {
    if (tcfg[3].n_cols != ORC_KECCAK_COLS) return -2;
    tcfg[3].air_id = ORC_AIR_KECCAK_F;
  }
  if (logic_air) {
    if (tcfg[5].n_cols != ORC_LOGIC_COLS) return -2;
    tcfg[5].air_id = ORC_AIR_LOGIC;
  }
  if (memory_air) {
    if (tcfg[6].n_cols != ORC_MEMORY_COLS) return -2;
    tcfg[6].air_id = ORC_AIR_MEMORY;
  }
  if (arithmetic_air) {
    if (tcfg[0].n_cols != ORC_ARITHMETIC_COLS) return -2;
    tcfg[0].air_id = ORC_AIR_ARITHMETIC;
  }
  if (mul_air) {
    if (tcfg[0].n_cols != ORC_ARITHMETIC_MUL_COLS) return -2;
    tcfg[0].air_id = ORC_AIR_ARITHMETIC_MUL;
  }
  if (byte_packing_air) {
    if (tcfg[1].n_cols != ORC_BYTE_PACKING_COLS) return -2;
    tcfg[1].air_id = ORC_AIR_BYTE_PACKING;
  }
  if (sponge_air) {
    if (tcfg[4].n_cols != ORC_KECCAK_SPONGE_COLS) return -2;
    tcfg[4].air_id = ORC_AIR_KECCAK_SPONGE;
  }
  if (wit) {
    const int has_air[NUM_TABLES] = {arithmetic_air || mul_air, byte_packing_air, 0, keccak_air, sponge_air, logic_air, memory_air};
    for (int t = 0; t < NUM_TABLES; t++) {
      if (!wit->items[t]) continue;
      const size_t rows = (size_t)1 << tcfg[t].log_n;
      if (!has_air[t] || wit->n[t] > (t == 3 ? (rows + 23) / 24 : rows)) return -3;
    }
  }
  pv[0] = I[3]; pv[1] = I[3] + (dummy ? 0 : 1); pv[2] = I[4]; pv[3] = I[5];
  memcpy(pv + 4, I + 6, 32);
  gl_t in6[6] = {I[6], I[7], I[8], I[9], gl_canon(I[10]), gl_canon(I[3])};
  if (dummy) memcpy(pv + 8, I + 6, 32);
  else orc_hash_no_pad(in6, 6, pv + 8);
  pv[12] = I[2];
  for (int i = 0; i < PV_WORDS; i++) pv[i] = gl_canon(pv[i]);

  gl_t* trace[NUM_TABLES];
  orc_committed* tc[NUM_TABLES];
  orc_challenger ch;
  orc_ch_init(&ch);
  /* The lookup keccak_sponge -> keccak_f (ctl.c) makes two real tables one statement.  Seeded tables are drawn so that
   * it holds: the sponge table absorbs blocks only in rows whose permutation the Keccak-f table holds in full, and
   * permutation p of the Keccak-f table is the one sponge row p asks for.  Tables given by the caller are taken as
   * they are (and refused below if they disagree). */
  const int lookup_kf = keccak_air && sponge_air;
  const size_t kf_full_perms = ((size_t)1 << tcfg[3].log_n) / 24;
  /* keccak_sponge -> logic (ctl.c): rows 5 p + m of the logic table are the five XORs of sponge row p (where it absorbs a
   * block; a row without an operation where it does not), for the sponge rows the logic table has room for; the caller's
   * or the seeded operations follow.  The seeded sponge table absorbs no block beyond those rows. */
  const int lookup_sl = sponge_air && logic_air;
  const size_t sl_covered = lookup_sl ? (((size_t)1 << tcfg[4].log_n) < ((size_t)1 << tcfg[5].log_n) / 5 ? ((size_t)1 << tcfg[4].log_n)
                                                                                                       : ((size_t)1 << tcfg[5].log_n) / 5) : 0;
  if (lookup_sl && wit && wit->items[5] && wit->n[5] > ((size_t)1 << tcfg[5].log_n) - 5 * sl_covered) return -2;
  /* byte_packing -> memory (ctl.c): a memory table that is not given is the log of the byte-packing table's words --
   * per packing row the write that put the word at its address (timestamp 1) and the operation the row looks up --,
   * then re-reads of the last address up to the table's height */
  const int lookup_bm = byte_packing_air && memory_air;
  if (lookup_bm && wit && wit->items[1] && !wit->items[6]) return -2; /* given sequences need their log */
  if (lookup_bm && !(wit && wit->items[6]) && tcfg[6].log_n < tcfg[1].log_n + 1) return -2;
  static const int order[NUM_TABLES] = {4, 0, 1, 2, 3, 5, 6}; /* the sponge table first: the Keccak-f table reads it */
  for (int oi = 0; oi < NUM_TABLES; oi++) {
    const int t = order[oi];
    size_t n = (size_t)1 << tcfg[t].log_n;
    trace[t] = (gl_t*)malloc(tcfg[t].n_cols * n * sizeof(gl_t));
    if (t == 5 && lookup_sl) {
      const uint64_t seed = I[10] ^ splitmix64(t + 1);
      const size_t ns = (size_t)1 << tcfg[4].log_n;
      const int given = wit && wit->items[5];
      uint64_t* in = (uint64_t*)calloc(n * 9, 8);
      for (size_t i = 0; i < n; i++) {
        uint64_t* o = in + i * 9;
        if (i < 5 * sl_covered) {
          const size_t p = i / 5, m = i % 5;
          if (!(trace[4][0 * ns + p] || trace[4][1 * ns + p])) continue; /* no operation */
          o[0] = 3;
          for (int w = 0; w < 4; w++)
            for (int z = 0; z < 64; z++) {
              const size_t bit = 256 * m + 64 * (size_t)w + z;
              if (bit >= 1088) break;
              o[1 + w] |= (trace[4][(1226 + bit) * ns + p] & 1) << z; /* the rate before the block */
              o[5 + w] |= (trace[4][(138 + bit) * ns + p] & 1) << z;  /* the block */
            }
        } else if (given) {
          const size_t k = i - 5 * sl_covered;
          if (k < wit->n[5]) memcpy(o, wit->items[5] + k * 9, 72);
        } else {
          o[0] = splitmix64(seed ^ (0xFFULL << 32) ^ i) & 3;
          for (int w = 0; w < 8; w++) o[1 + w] = splitmix64(seed ^ ((uint64_t)(1 + w) << 32) ^ i);
        }
      }
      orc_logic_trace(0, in, tcfg[t].log_n, trace[t]);
      free(in);
    } else if (wit && wit->items[t] && tcfg[t].air_id != ORC_AIR_SYNTHETIC) {
      uint64_t* in = padded_items(t, n, wit->items[t], wit->n[t]);
      if (t == 3) orc_keccak_trace(0, in, tcfg[t].log_n, trace[t]);
      else if (t == 5) orc_logic_trace(0, in, tcfg[t].log_n, trace[t]);
      else if (t == 6) orc_memory_trace(0, in, tcfg[t].log_n, trace[t]);
      else if (t == 0 && mul_air) orc_arithmetic_mul_trace(0, in, tcfg[t].log_n, trace[t]);
      else if (t == 0) orc_arithmetic_trace(0, in, tcfg[t].log_n, trace[t]);
      else if (t == 4) orc_keccak_sponge_trace(0, in, tcfg[t].log_n, trace[t]);
      else orc_byte_packing_trace(0, in, tcfg[t].log_n, trace[t]);
      free(in);
    } else if (tcfg[t].air_id == ORC_AIR_KECCAK_F && lookup_kf) {
      /* permutation p: what sponge row p asks for, if it absorbs a block; else drawn from the seed as usual */
      const uint64_t seed = I[10] ^ splitmix64(t + 1);
      const size_t n_perm = (n + 23) / 24, ns = (size_t)1 << tcfg[4].log_n;
      uint64_t* in = (uint64_t*)malloc(n_perm * 25 * 8);
      for (size_t p = 0; p < n_perm; p++) {
        const int asked = p < ns && (trace[4][0 * ns + p] || trace[4][1 * ns + p]);
        for (int l = 0; l < 25; l++) {
          if (asked) {
            const size_t col = l < 17 ? 2330 + 2 * (size_t)l : 2314 + 2 * (size_t)(l - 17); /* xored rate | capacity limbs */
            in[p * 25 + l] = trace[4][col * ns + p] | (trace[4][(col + 1) * ns + p] << 32);
          } else {
            in[p * 25 + l] = splitmix64(seed ^ ((uint64_t)l << 32) ^ p);
          }
        }
      }
      orc_keccak_trace(0, in, tcfg[t].log_n, trace[t]);
      free(in);
    } else if (tcfg[t].air_id == ORC_AIR_MEMORY && lookup_bm) {
      const size_t np = (size_t)1 << tcfg[1].log_n;
      const gl_t* pk = trace[1];
      uint64_t* in = (uint64_t*)malloc(n * 11 * 8);
      for (size_t i = 0; i < n; i++) {
        const size_t r = i / 2 < np ? i / 2 : np - 1;
        uint64_t* o = in + i * 11;
        o[1] = pk[297 * np + r];
        for (int k = 0; k < 8; k++) o[3 + k] = pk[(size_t)(289 + k) * np + r];
        if (i / 2 >= np) { o[0] = 1; o[2] = pk[298 * np + r] + (i - 2 * np + 1); }   /* a re-read of the last address */
        else if (i % 2 == 0) { o[0] = 0; o[2] = 1; }                                  /* the word is put there */
        else { o[0] = pk[0 * np + r]; o[2] = pk[298 * np + r]; }                      /* the operation the packing row names */
      }
      orc_memory_trace(0, in, tcfg[t].log_n, trace[t]);
      free(in);
    } else if (tcfg[t].air_id == ORC_AIR_KECCAK_F) orc_keccak_trace(I[10] ^ splitmix64(t + 1), NULL, tcfg[t].log_n, trace[t]);
    else if (tcfg[t].air_id == ORC_AIR_LOGIC) orc_logic_trace(I[10] ^ splitmix64(t + 1), NULL, tcfg[t].log_n, trace[t]);
    else if (tcfg[t].air_id == ORC_AIR_MEMORY) orc_memory_trace(I[10] ^ splitmix64(t + 1), NULL, tcfg[t].log_n, trace[t]);
    else if (tcfg[t].air_id == ORC_AIR_ARITHMETIC) orc_arithmetic_trace(I[10] ^ splitmix64(t + 1), NULL, tcfg[t].log_n, trace[t]);
    else if (tcfg[t].air_id == ORC_AIR_ARITHMETIC_MUL) orc_arithmetic_mul_trace(I[10] ^ splitmix64(t + 1), NULL, tcfg[t].log_n, trace[t]);
    else if (tcfg[t].air_id == ORC_AIR_BYTE_PACKING) orc_byte_packing_trace(I[10] ^ splitmix64(t + 1), NULL, tcfg[t].log_n, trace[t]);
    else if (tcfg[t].air_id == ORC_AIR_KECCAK_SPONGE)
      orc_keccak_sponge_trace_limit(I[10] ^ splitmix64(t + 1), NULL, tcfg[t].log_n,
                                    (lookup_kf ? kf_full_perms : (size_t)-1) < (lookup_sl ? sl_covered : (size_t)-1)
                                        ? (lookup_kf ? kf_full_perms : (size_t)-1) : (lookup_sl ? sl_covered : (size_t)-1), trace[t]);
    else orc_synth_trace(I[10] ^ splitmix64(t + 1), &tcfg[t], NULL, trace[t]);
  }
  /* the filter columns of the two looked tables are part of their traces: set BEFORE the traces are committed */
  if (lookup_kf) { /* the Keccak-f table exposes the permutations the sponge rows ask for, row p <-> permutation p */
    const size_t ns = (size_t)1 << tcfg[4].log_n;
    uint8_t* exposed = (uint8_t*)malloc(ns);
    for (size_t p = 0; p < ns; p++) exposed[p] = trace[4][p] || trace[4][ns + p];
    orc_ctl_set_filter(ORC_AIR_KECCAK_F, trace[3], tcfg[3].log_n, exposed, ns);
    free(exposed);
  }
  if (lookup_sl) { /* the logic table exposes rows 5 p + m of the sponge rows p that absorb a block */
    const size_t ns = (size_t)1 << tcfg[4].log_n, nl = (size_t)1 << tcfg[5].log_n;
    uint8_t* exposed = (uint8_t*)calloc(nl, 1);
    for (size_t i = 0; i < 5 * sl_covered; i++) exposed[i] = trace[4][i / 5] || trace[4][ns + i / 5];
    orc_ctl_set_filter(ORC_AIR_LOGIC, trace[5], tcfg[5].log_n, exposed, nl);
    free(exposed);
  }
  if (lookup_bm) { /* the memory table exposes the operations the byte-packing rows name */
    const size_t np = (size_t)1 << tcfg[1].log_n, nm = (size_t)1 << tcfg[6].log_n;
    uint8_t* exposed = (uint8_t*)calloc(nm, 1);
    for (size_t r = 0; r < np; r++) {
      int moves = 0;
      for (int j = 0; j < 32; j++) moves |= trace[1][(size_t)(1 + j) * np + r] != 0;
      if (!moves) continue;
      const gl_t a = trace[1][297 * np + r], ts = trace[1][298 * np + r];
      for (size_t i = 0; i < nm; i++)
        if (trace[6][1 * nm + i] == a && trace[6][2 * nm + i] == ts) { exposed[i] = 1; break; }
    }
    orc_ctl_set_filter(ORC_AIR_MEMORY, trace[6], tcfg[6].log_n, exposed, nm);
    free(exposed);
  }
  for (int t = 0; t < NUM_TABLES; t++) {
    tc[t] = orc_commit_values(trace[t], tcfg[t].log_n, tcfg[t].n_cols, tcfg[t].rate_bits, tcfg[t].cap_height);
    orc_ch_observe_many(&ch, orc_committed_cap(tc[t]), (size_t)4 << tcfg[t].cap_height);
  }
  orc_ch_observe_many(&ch, pv, PV_WORDS);
  for (int i = 0; i < 4; i++) ctl[i] = orc_ch_challenge(&ch);
  int rc = 0;
  for (int t = 0; t < NUM_TABLES; t++) proofs[t] = NULL;
  for (int t = 0; t < NUM_TABLES && !rc; t++) {
    proofs[t] = (gl_t*)malloc(orc_proof_words(&tcfg[t]) * sizeof(gl_t));
    rc = orc_stark_prove(&tcfg[t], NULL, tc[t], trace[t], ctl, &ch, proofs[t]);
  }
  for (int t = 0; t < NUM_TABLES; t++) { orc_committed_free(tc[t]); free(trace[t]); }
  if (!rc && g_prover_checks_lookups) rc = orc_pg_check_lookups(tcfg, (const gl_t* const*)proofs);
  if (rc) for (int t = 0; t < NUM_TABLES; t++) { free(proofs[t]); proofs[t] = NULL; }
  return rc;
}

/* Offsets of the first-row openings of the auxiliary columns inside a table proof (stark.c layout): after the 16-word
 * header, three caps, the openings at zeta (all columns + quotient chunks) and at g*zeta (trace + aux). */
static size_t open_first_offset(const orc_stark_cfg* c) {
  const size_t cap = (size_t)4 << c->cap_height, A = orc_cfg_n_aux(c), Q = orc_cfg_n_quot(c);
  return 16 + 3 * cap + 2 * ((size_t)c->n_const + c->n_cols + A + Q) + 2 * ((size_t)c->n_cols + A);
}
/* The lookups between the tables of one transaction (ctl.c): the sponge table's running product and the Keccak-f
 * table's agree at the first row, for both challenge sets.  -11: they do not. */
int orc_pg_check_lookups(const orc_stark_cfg tcfg[NUM_TABLES], const gl_t* const proofs[NUM_TABLES]) {
  if (tcfg[3].air_id == ORC_AIR_KECCAK_F && tcfg[4].air_id == ORC_AIR_KECCAK_SPONGE) {
    const gl_t* looking = proofs[4] + open_first_offset(&tcfg[4]); /* z_0, z_1 are its aux columns 0, 1 */
    const gl_t* looked = proofs[3] + open_first_offset(&tcfg[3]);  /* ... and columns 2, 3 here (after h_0, h_1) */
    for (int c = 0; c < 2; c++)
      if (looking[2 * c] != looked[2 * (2 + c)] || looking[2 * c + 1] != looked[2 * (2 + c) + 1]) return -11;
  }
  if (tcfg[4].air_id == ORC_AIR_KECCAK_SPONGE && tcfg[5].air_id == ORC_AIR_LOGIC) {
    /* the sponge table's ten products into the logic table (columns 2 + 2 m + c), multiplied over m, against the logic
     * table's z_c: first-row openings are extension elements (c0, c1) */
    const gl_t* looking = proofs[4] + open_first_offset(&tcfg[4]);
    const gl_t* looked = proofs[5] + open_first_offset(&tcfg[5]);
    for (int c = 0; c < 2; c++) {
      gl2_t prod = gl2_from(1);
      for (int m = 0; m < 5; m++) {
        const gl_t* v = looking + 2 * (2 + 2 * m + c);
        gl2_t x; x.c0 = v[0]; x.c1 = v[1];
        prod = gl2_mul(prod, x);
      }
      if (prod.c0 != looked[2 * c] || prod.c1 != looked[2 * c + 1]) return -13;
    }
  }
  if (tcfg[1].air_id == ORC_AIR_BYTE_PACKING && tcfg[6].air_id == ORC_AIR_MEMORY) {
    const gl_t* looking = proofs[1] + open_first_offset(&tcfg[1]); /* z_0, z_1 are its aux columns 0, 1 */
    const gl_t* looked = proofs[6] + open_first_offset(&tcfg[6]);  /* ... and columns 0, 1 here too (the filter is a trace column) */
    for (int c = 0; c < 2; c++)
      if (looking[2 * c] != looked[2 * c] || looking[2 * c + 1] != looked[2 * c + 1]) return -12;
  }
  return 0;
}

#define TABLES_MAGIC 0x534C424154475042ULL /* "BPGTABLS" */
/* the table proofs alone, as libbpg's bp_generate_txn_table_proofs emits them */
static int pg_txn_tables(orc_pg_state* s, const uint64_t* I, const pg_witness* wit, gl_t** out, size_t* out_words) {
  orc_stark_cfg tcfg[NUM_TABLES];
  gl_t pv[PV_WORDS], ctl[4], *proofs[NUM_TABLES];
  int rc = pg_tables(s, I, wit, tcfg, pv, ctl, proofs);
  if (rc) return rc;
  size_t words = 2 + PV_WORDS + 4;
  for (int t = 0; t < NUM_TABLES; t++) words += 4 + orc_proof_words(&tcfg[t]);
  gl_t* o = (gl_t*)malloc(words * sizeof(gl_t));
  size_t off = 0;
  o[off++] = TABLES_MAGIC; o[off++] = NUM_TABLES;
  memcpy(o + off, pv, sizeof(pv)); off += PV_WORDS;
  memcpy(o + off, ctl, 32); off += 4;
  for (int t = 0; t < NUM_TABLES; t++) {
    const size_t pw = orc_proof_words(&tcfg[t]);
    o[off++] = tcfg[t].air_id; o[off++] = tcfg[t].log_n; o[off++] = tcfg[t].n_cols; o[off++] = pw;
    memcpy(o + off, proofs[t], pw * 8); off += pw;
    free(proofs[t]);
  }
  *out = o; *out_words = words;
  return 0;
}
int orc_pg_txn_tables(orc_pg_state* s, const uint64_t* I, const uint64_t* const items[NUM_TABLES], const size_t n[NUM_TABLES],
                      const int given[NUM_TABLES], gl_t** out, size_t* out_words) {
  static const uint64_t none = 0;
  pg_witness w;
  memset(&w, 0, sizeof(w));
  int any = 0;
  for (int t = 0; t < NUM_TABLES; t++)
    if (given && given[t] && WIT_WORDS[t]) {
      w.items[t] = items[t] ? items[t] : &none;
      w.n[t] = n[t];
      any = 1;
    }
  return pg_txn_tables(s, I, any ? &w : NULL, out, out_words);
}
/* upstream's verify_proof(all_stark, all_proof, config): every table proof against the shared transcript, then the
 * lookups between the tables.  0 = accept; the first failing table t gives -(100 + 20 t) + (the table verifier's code). */
int orc_pg_verify_tables(const orc_pg_config* cfg, const gl_t* w, size_t words) {
  if (words < 2 + PV_WORDS + 4 || w[0] != TABLES_MAGIC || w[1] != NUM_TABLES) return -2;
  const gl_t *pv = w + 2, *ctl_in = pv + PV_WORDS;
  orc_stark_cfg tcfg[NUM_TABLES];
  const gl_t* proofs[NUM_TABLES];
  size_t off = 2 + PV_WORDS + 4;
  static const uint32_t table_air[NUM_TABLES] = {ORC_AIR_ARITHMETIC, ORC_AIR_BYTE_PACKING, 0xFFFFFFFFu, ORC_AIR_KECCAK_F,
                                                 ORC_AIR_KECCAK_SPONGE, ORC_AIR_LOGIC, ORC_AIR_MEMORY};
  for (int t = 0; t < NUM_TABLES; t++) {
    if (off + 4 > words) return -2;
    if (w[off] != ORC_AIR_SYNTHETIC && w[off] != table_air[t] && !(t == 0 && w[off] == ORC_AIR_ARITHMETIC_MUL)) return -5;
    if (w[off + 1] > 30 || w[off + 2] > 65536) return -2;
    tcfg[t] = table_cfg_of(cfg, (uint32_t)w[off + 1], (uint32_t)w[off + 2]);
    tcfg[t].air_id = (uint32_t)w[off];
    if (w[off + 3] != orc_proof_words(&tcfg[t]) || off + 4 + w[off + 3] > words) return -2;
    proofs[t] = w + off + 4;
    off += 4 + w[off + 3];
  }
  if (off != words) return -2;
  orc_challenger ch;
  orc_ch_init(&ch);
  for (int t = 0; t < NUM_TABLES; t++) orc_ch_observe_many(&ch, proofs[t] + 16, (size_t)4 << tcfg[t].cap_height);
  orc_ch_observe_many(&ch, pv, PV_WORDS);
  gl_t ctl[4];
  for (int i = 0; i < 4; i++) {
    ctl[i] = orc_ch_challenge(&ch);
    if (ctl[i] != ctl_in[i]) return -6;
  }
  for (int t = 0; t < NUM_TABLES; t++) {
    int rc = orc_stark_verify(&tcfg[t], NULL, ctl, &ch, proofs[t]);
    if (rc) return -(100 + 20 * t) + rc;
  }
  return orc_pg_check_lookups(tcfg, proofs);
}

static int pg_txn(orc_pg_state* s, const uint64_t* I, const pg_witness* wit, gl_t** out, size_t* out_words) {
  const orc_pg_config* cfg = &s->cfg;
  orc_stark_cfg tcfg[NUM_TABLES];
  gl_t pv[PV_WORDS], ctl[4], *tproofs[NUM_TABLES];
  int rc = pg_tables(s, I, wit, tcfg, pv, ctl, tproofs);
  if (rc) return rc;
  /* per chain: the digest of its newest proof and that proof's first trace opening (leaf digest, cap entry, path) */
  gl_t digest[NUM_TABLES][4], leaf_cap[NUM_TABLES][8], *path[NUM_TABLES];
  /* (a recursion-shaped child's path is followed by its opened row: the shrink and root circuits hash it when they have room) */
  const size_t rec_sib_words = 1 + 4 * (size_t)(s->rec.log_n + s->rec.rate_bits - s->rec.cap_height);
  const size_t rec_path_words = rec_sib_words + s->rec.n_cols;
  for (int t = 0; t < NUM_TABLES; t++) {
    const size_t table_path_words = 1 + 4 * (size_t)(tcfg[t].log_n + tcfg[t].rate_bits - tcfg[t].cap_height);
    path[t] = (gl_t*)malloc((table_path_words > rec_path_words ? table_path_words : rec_path_words) * sizeof(gl_t));
    orc_proof_digest(&tcfg[t], tproofs[t], digest[t]);
    orc_proof_first_query_path(&tcfg[t], tproofs[t], leaf_cap[t], leaf_cap[t] + 4, path[t]);
    free(tproofs[t]);
  }
  size_t sw = orc_proof_words(&s->rec);
  gl_t* proof = (gl_t*)malloc(sw * sizeof(gl_t));
  for (int t = 0; t < NUM_TABLES && !rc; t++) {
    for (uint32_t depth = 0; depth < cfg->shrink_depth && !rc; depth++) {
      circuit_t* circ = depth == 0 ? table_circuit(s, t, tcfg[t].log_n) : shrink_circuit(s, t);
      gl_t pi[CHAIN_PI] = {digest[t][0], digest[t][1], digest[t][2], digest[t][3], (gl_t)t, depth};
      memcpy(pi + CHAIN_PATH_AT, leaf_cap[t], sizeof(leaf_cap[t]));
      rc = rec_prove(s, circ, pi, CHAIN_PI, path[t], proof);
      if (!rc) {
        orc_proof_digest(&s->rec, proof, digest[t]);
        orc_proof_first_query_path(&s->rec, proof, leaf_cap[t], leaf_cap[t] + 4, path[t]);
        orc_proof_first_query_row(&s->rec, proof, path[t] + rec_sib_words);
      }
    }
  }
  if (!rc) {
    const size_t root_path_words = rec_sib_words + special_circuit(s, 0)->leaf_len;
    gl_t pi[ROOT_PI], *paths = (gl_t*)malloc(NUM_TABLES * root_path_words * sizeof(gl_t));
    for (int t = 0; t < NUM_TABLES; t++) {
      memcpy(pi + 4 * t, digest[t], 32);
      memcpy(pi + ROOT_PATHS_AT + 8 * t, leaf_cap[t], sizeof(leaf_cap[t]));
      memcpy(paths + t * root_path_words, path[t], root_path_words * sizeof(gl_t));
    }
    memcpy(pi + ROOT_PATHS_AT + 8 * NUM_TABLES, pv, sizeof(pv));
    rc = rec_prove(s, special_circuit(s, 0), pi, ROOT_PI, paths, proof);
    if (!rc) *out = emit_box(0, pi, ROOT_PI, proof, sw, out_words);
    free(paths);
  }
  for (int t = 0; t < NUM_TABLES; t++) free(path[t]);
  free(proof);
  return rc;
}

typedef struct { uint64_t kind, n_pi; const gl_t *pi, *pv, *stark; } box_t;
static int parse_box(orc_pg_state* s, const gl_t* w, size_t words, box_t* b) {
  if (words < BOX_HDR + PV_WORDS || w[0] != BOX_MAGIC || w[1] > 2 || w[2] < PV_WORDS || w[2] > 104) return -2;
  b->kind = w[1]; b->n_pi = w[2];
  if (words != BOX_HDR + b->n_pi + orc_proof_words(&s->rec)) return -2;
  b->pi = w + BOX_HDR; b->pv = b->pi + b->n_pi - PV_WORDS; b->stark = b->pi + b->n_pi;
  return 0;
}

/* generate_agg_proof (proof_gen.rs:61-79) */
int orc_pg_agg(orc_pg_state* s, const gl_t* lhs, size_t lw, int lhs_is_agg, const gl_t* rhs, size_t rw,
               int rhs_is_agg, gl_t** out, size_t* out_words) {
  box_t L, R;
  if (parse_box(s, lhs, lw, &L) || parse_box(s, rhs, rw, &R)) return -2;
  if ((L.kind == 1) != (lhs_is_agg != 0) || (R.kind == 1) != (rhs_is_agg != 0) || L.kind > 1 || R.kind > 1) return -2;
  if (L.pv[1] != R.pv[0] || L.pv[3] != R.pv[2] || memcmp(L.pv + 8, R.pv + 4, 32) || L.pv[12] != R.pv[12]) return -2;
  gl_t pi[AGG_PI];
  orc_proof_digest(&s->rec, L.stark, pi);
  orc_proof_digest(&s->rec, R.stark, pi + 4);
  pi[8] = lhs_is_agg != 0; pi[9] = rhs_is_agg != 0;
  const unsigned agg_leaf = special_circuit(s, 1)->leaf_len;
  const size_t sib_words = 1 + 4 * (size_t)(s->rec.log_n + s->rec.rate_bits - s->rec.cap_height), path_words = sib_words + agg_leaf;
  gl_t* paths = (gl_t*)malloc(2 * path_words * sizeof(gl_t));
  orc_proof_first_query_path(&s->rec, L.stark, pi + AGG_PATHS_AT, pi + AGG_PATHS_AT + 4, paths);
  orc_proof_first_query_path(&s->rec, R.stark, pi + AGG_PATHS_AT + 8, pi + AGG_PATHS_AT + 12, paths + path_words);
  if (agg_leaf) { /* the opened rows themselves, hashed in-circuit */
    orc_proof_first_query_row(&s->rec, L.stark, paths + sib_words);
    orc_proof_first_query_row(&s->rec, R.stark, paths + path_words + sib_words);
  }
  gl_t* pv = pi + AGG_PATHS_AT + 16;
  pv[0] = L.pv[0]; pv[1] = R.pv[1]; pv[2] = L.pv[2]; pv[3] = R.pv[3];
  memcpy(pv + 4, L.pv + 4, 32); memcpy(pv + 8, R.pv + 8, 32);
  pv[12] = L.pv[12];
  size_t sw = orc_proof_words(&s->rec);
  gl_t* proof = (gl_t*)malloc(sw * sizeof(gl_t));
  int rc = rec_prove(s, special_circuit(s, 1), pi, AGG_PI, paths, proof);
  if (!rc) *out = emit_box(1, pi, AGG_PI, proof, sw, out_words);
  free(proof);
  free(paths);
  return rc;
}

/* generate_block_proof (proof_gen.rs:85-110) */
int orc_pg_block(orc_pg_state* s, const gl_t* parent, size_t pw, const gl_t* agg, size_t aw, gl_t** out,
                 size_t* out_words) {
  box_t A, Pb;
  if (parse_box(s, agg, aw, &A) || A.kind != 1) return -2;
  gl_t pi[BLOCK_PI];
  memset(pi, 0, sizeof(pi));
  if (parent) {
    if (parse_box(s, parent, pw, &Pb) || Pb.kind != 2 || Pb.pv[12] + 1 != A.pv[12]) return -2;
    orc_proof_digest(&s->rec, Pb.stark, pi);
    pi[8] = 1;
  }
  orc_proof_digest(&s->rec, A.stark, pi + 4);
  const unsigned blk_leaf = special_circuit(s, 2)->leaf_len;
  const size_t blk_sib_words = 1 + 4 * (size_t)(s->rec.log_n + s->rec.rate_bits - s->rec.cap_height);
  gl_t* path = (gl_t*)malloc((blk_sib_words + blk_leaf) * sizeof(gl_t));
  orc_proof_first_query_path(&s->rec, A.stark, pi + BLOCK_PATH_AT, pi + BLOCK_PATH_AT + 4, path);
  if (blk_leaf) orc_proof_first_query_row(&s->rec, A.stark, path + blk_sib_words);
  memcpy(pi + BLOCK_PATH_AT + 8, A.pv, PV_WORDS * 8);
  size_t sw = orc_proof_words(&s->rec);
  gl_t* proof = (gl_t*)malloc(sw * sizeof(gl_t));
  int rc = rec_prove(s, special_circuit(s, 2), pi, BLOCK_PI, path, proof);
  if (!rc) *out = emit_box(2, pi, BLOCK_PI, proof, sw, out_words);
  free(proof);
  free(path);
  return rc;
}

/* VerifierState::verify (verifier_state.rs:56-71) */
int orc_pg_verify(orc_pg_state* s, const gl_t* w, size_t words) {
  box_t b;
  if (parse_box(s, w, words, &b)) return -2;
  if (w[3] != CIRCUIT_ROOT + b.kind) return -5;
  circuit_t* circ = special_circuit(s, (int)b.kind);
  gl_t pi_hash[4];
  orc_hash_no_pad(b.pi, b.n_pi, pi_hash);
  orc_challenger ch;
  orc_ch_init(&ch);
  orc_ch_observe_many(&ch, circ->digest, 4);
  orc_ch_observe_many(&ch, pi_hash, 4);
  size_t cap_words = (size_t)4 << s->rec.cap_height;
  orc_ch_observe_many(&ch, b.stark + 16, cap_words); /* trace cap follows the 16-word header */
  gl_t ctl[4];
  for (int i = 0; i < 4; i++) ctl[i] = orc_ch_challenge(&ch);
  orc_stark_cfg rcfg = s->rec;
  if (rcfg.air_id == ORC_AIR_PLONK) memcpy(rcfg.pub, pi_hash, sizeof(rcfg.pub));
  return orc_stark_verify(&rcfg, orc_committed_cap(circ->consts), ctl, &ch, b.stark);
}

/* cap of special circuit k (0 root, 1 agg, 2 block): what a light verifier keeps */
void orc_pg_circuit_cap(orc_pg_state* s, int k, gl_t* out) {
  circuit_t* c = special_circuit(s, k);
  memcpy(out, orc_committed_cap(c->consts), ((size_t)4 << s->rec.cap_height) * sizeof(gl_t));
}

void orc_free(void* p) { free(p); }
