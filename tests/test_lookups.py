"""Cross-table lookups (CPU): the tables of a transaction that are proven with their AIRs form ONE statement
(proof_gen.rs:44-52 proves all tables in one call; upstream ties them with plonky2_evm's cross-table lookups).  Built
here: keccak_sponge -> keccak_f, keccak_sponge -> logic and byte_packing -> memory (csrc/air.hpp namespace ctl, oracle/ctl.c, AIRS.md section 3).  The oracle proves a
transaction's seven tables; its own verifier and the product's CPU verifier (independent statements of the lookup
constraints) both accept; tables that are each valid alone but disagree with each other are rejected by both."""
import numpy as np
import pytest

from pg_common import LOG_N, SMALL, WIDTH, ir_words

REAL_WIDTH = {0: 309, 1: 299, 3: 2431, 4: 2414, 5: 524, 6: 45}
REAL_FLAG = {3: 0x100, 5: 0x200, 6: 0x400, 0: 0x800, 1: 0x1000, 4: 0x2000}


def real_ir(tables, seed=0x5EED0C71, log_n=LOG_N):
    w = [REAL_WIDTH[t] if t in tables else WIDTH[t] for t in range(7)]
    ir = ir_words(9, 0, seed, log_n=log_n, width=tuple(w))
    for t in tables:
        ir[1] |= REAL_FLAG[t]
    return ir


@pytest.fixture(scope="module")
def o_state(oracle):
    return oracle.PgState(**SMALL)


@pytest.fixture(scope="module")
def product_cfg():
    from proof_protocol_decoder_amd import proof_gen as pg
    b = pg.ProverStateBuilder()
    for t, name in enumerate(pg.TABLES):
        getattr(b, "set_%s_circuit_size" % name)(range(SMALL["table_log_lo"][t], SMALL["table_log_hi"][t]))
    b.set(**{k: v for k, v in SMALL.items() if not k.startswith("table_")})
    return pg, b.cfg


def table_slices(tp):
    """[(air_id, log_n, n_cols, first word, n words)] of a "BPGTABLS" container."""
    out, off = [], 2 + 13 + 4
    for _ in range(7):
        air, log_n, n_cols, pw = (int(x) for x in tp[off:off + 4])
        out.append((air, log_n, n_cols, off + 4, pw))
        off += 4 + pw
    assert off == tp.size
    return out


def first_row_openings(oracle, tp, t):
    """the auxiliary columns of table t opened at the first row, as (c0, c1) pairs"""
    air, log_n, n_cols, first, _ = table_slices(tp)[t]
    a = int(oracle.lib().orc_ctl_n_aux(air, n_cols))
    q, cap = 4, 4 << SMALL["stark_cap_height"]
    off = first + 16 + 3 * cap + 2 * (n_cols + a + q) + 2 * (n_cols + a)
    return tp[off:off + 2 * a].reshape(a, 2)


def test_seeded_tables_of_a_transaction_are_one_statement(oracle, o_state, product_cfg):
    pg, cfg = product_cfg
    tp = o_state.txn_tables(real_ir({0, 1, 3, 4, 5, 6}))
    assert o_state.verify_tables(tp) == 0
    pg.verify_txn_table_proofs(cfg, tp.tobytes())
    # the lookup is not vacuous: the sponge table asks for permutations (its product is not 1) and the Keccak-f table's
    # exposed product is the same value, for both challenge sets
    looking, looked = first_row_openings(oracle, tp, 4), first_row_openings(oracle, tp, 3)
    assert (looking[0] == looked[2]).all() and (looking[1] == looked[3]).all()
    assert tuple(looking[0]) != (1, 0) and tuple(looking[0]) != tuple(looking[1])
    # the same for byte_packing -> memory: the packing rows that move a word name operations the memory table exposes
    packing, memory = first_row_openings(oracle, tp, 1), first_row_openings(oracle, tp, 6)
    assert (packing[0] == memory[0]).all() and (packing[1] == memory[1]).all()
    assert tuple(packing[0]) != (1, 0) and tuple(packing[0]) != tuple(packing[1])
    # keccak_sponge -> logic: the XORs of every absorbed block are operations of the logic table -- per challenge set the
    # product of the sponge table's five columns (limb groups m = 0..4: aux columns 2 + 2 m + c) is the logic table's z_c
    logic = first_row_openings(oracle, tp, 5)
    assert looking.shape == (12, 2) and logic.shape == (2, 2)
    for c in range(2):
        prod = (1, 0)
        for m in range(5):
            prod = ext_mul(prod, tuple(int(x) for x in looking[2 + 2 * m + c]))
        assert prod == tuple(int(x) for x in logic[c]) and prod != (1, 0)
    # a table no lookup is built for carries the constant product
    assert (first_row_openings(oracle, tp, 0) == [[1, 0]]).all()
    # a real logic table next to a synthetic sponge table exposes nothing
    tp4 = o_state.txn_tables(real_ir({5}))
    assert o_state.verify_tables(tp4) == 0
    pg.verify_txn_table_proofs(cfg, tp4.tobytes())
    assert (first_row_openings(oracle, tp4, 5) == [[1, 0], [1, 0]]).all()
    # a real memory table next to a synthetic byte-packing table exposes nothing
    tp3 = o_state.txn_tables(real_ir({6}))
    assert o_state.verify_tables(tp3) == 0
    pg.verify_txn_table_proofs(cfg, tp3.tobytes())
    assert (first_row_openings(oracle, tp3, 6) == [[1, 0], [1, 0]]).all()
    # without a real sponge table the Keccak-f table exposes nothing, and nothing is compared
    tp2 = o_state.txn_tables(real_ir({3}))
    assert o_state.verify_tables(tp2) == 0
    pg.verify_txn_table_proofs(cfg, tp2.tobytes())
    assert (first_row_openings(oracle, tp2, 3)[2:] == [[1, 0], [1, 0]]).all()


P = 0xFFFFFFFF00000001


def ext_mul(a, b):
    """(a0 + a1 X)(b0 + b1 X) with X^2 = 7"""
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def test_sponge_rows_the_logic_table_has_no_operations_for_are_refused(oracle, o_state, product_cfg):
    """keccak_sponge -> logic alone (sponge and logic tables by their AIRs, the Keccak-f table synthetic): the logic
    table of the small configuration (2^6 rows) holds the XORs of twelve sponge rows; given sponge rows that absorb
    thirteen blocks are still a valid absorption, and the logic table is still a table of valid operations -- but one block's
    XORs are nobody's operations."""
    pg, cfg = product_cfg
    ir = real_ir({4, 5})
    rows, _ = sponge_and_keccak_work(oracle, [b"a" * 130, bytes(range(250)) * 4, b"xyz"])     # 1 + 8 + 1 = 10 blocks
    assert len(rows) == 10
    good = o_state.txn_tables(ir, witness={4: rows})
    assert o_state.verify_tables(good) == 0
    pg.verify_txn_table_proofs(cfg, good.tobytes())
    logic = first_row_openings(oracle, good, 5)
    assert tuple(int(x) for x in logic[0]) != (1, 0)
    # the caller's own logic operations follow the sponge table's in the logic table and do not disturb the lookup
    ops = [[3, 1, 2, 3, 4, 5, 6, 7, 8], [1, 9, 9, 9, 9, 7, 7, 7, 7]]
    both = o_state.txn_tables(ir, witness={4: rows, 5: ops})
    assert o_state.verify_tables(both) == 0
    pg.verify_txn_table_proofs(cfg, both.tobytes())
    rows13, _ = sponge_and_keccak_work(oracle, [bytes(range(250)) * 6, b"q"])                # 12 + 1 = 13 blocks
    assert len(rows13) == 13
    with pytest.raises(RuntimeError, match="-13"):          # the prover refuses to go on
        o_state.txn_tables(ir, witness={4: rows13})
    oracle.lib().orc_pg_set_prover_lookup_check(0)
    try:
        bad = o_state.txn_tables(ir, witness={4: rows13})
    finally:
        oracle.lib().orc_pg_set_prover_lookup_check(1)
    assert o_state.verify_tables(bad) == -13
    with pytest.raises(pg.ProofGenError, match="cross-table lookup keccak_sponge -> logic does not hold") as e:
        pg.verify_txn_table_proofs(cfg, bad.tobytes())
    assert e.value.code == -5


def sponge_and_keccak_work(oracle, messages):
    rows, perms = [], []
    for m in messages:   # (the rows against Keccak-256 itself: tests/test_keccak_sponge_air.py)
        _, r = oracle.keccak_sponge_rows(m)
        rows += [[int(w) for w in x] for x in r]
    for r in rows:   # permutation input = (state before) xor (block as absorbed), capacity untouched
        blk, st = r[2:19], r[19:44]
        perms.append([st[l] ^ blk[l] if l < 17 else st[l] for l in range(25)])
    return rows, perms


def test_given_tables_that_disagree_are_rejected_by_both_verifiers(oracle, o_state, product_cfg):
    pg, cfg = product_cfg
    msgs = [b"abc", bytes(range(200)), b""]
    rows, perms = sponge_and_keccak_work(oracle, msgs)
    ir = real_ir({3, 4})
    good = o_state.txn_tables(ir, witness={3: perms, 4: rows})
    assert o_state.verify_tables(good) == 0
    pg.verify_txn_table_proofs(cfg, good.tobytes())
    # one lane of one permutation's input differs: the Keccak-f table is still a table of valid permutations, the sponge
    # table still a valid absorption -- but what the one hashes is no longer what the other permutes
    bad_perms = [list(p) for p in perms]
    bad_perms[1][7] ^= 1 << 33
    with pytest.raises(RuntimeError, match="-11"):          # the prover refuses to go on
        o_state.txn_tables(ir, witness={3: bad_perms, 4: rows})
    oracle.lib().orc_pg_set_prover_lookup_check(0)
    try:
        bad = o_state.txn_tables(ir, witness={3: bad_perms, 4: rows})
    finally:
        oracle.lib().orc_pg_set_prover_lookup_check(1)
    assert o_state.verify_tables(bad) == -11
    with pytest.raises(pg.ProofGenError, match="cross-table lookup keccak_sponge -> keccak_f does not hold") as e:
        pg.verify_txn_table_proofs(cfg, bad.tobytes())
    assert e.value.code == -5
    # a sponge row the Keccak-f table has no permutation for (one message more than it permutes)
    rows2, _ = sponge_and_keccak_work(oracle, msgs + [b"one more"])
    oracle.lib().orc_pg_set_prover_lookup_check(0)
    try:
        bad2 = o_state.txn_tables(ir, witness={3: perms, 4: rows2})
    finally:
        oracle.lib().orc_pg_set_prover_lookup_check(1)
    assert o_state.verify_tables(bad2) == -11
    with pytest.raises(pg.ProofGenError, match="cross-table lookup"):
        pg.verify_txn_table_proofs(cfg, bad2.tobytes())


def test_the_filter_of_a_looked_table_is_committed_before_the_lookup_challenges(oracle, o_state, product_cfg):
    """which rows a looked table exposes (its filter g) was an AUXILIARY column, committed after the lookup
    challenges were drawn -- a prover could pick the exposed subset knowing beta and gamma.  It is a TRACE column now
    (Keccak-f: column 2430, memory: column 44): two provers whose Keccak-f tables hold the SAME permutations but expose
    different subsets commit to different trace caps, hence draw different challenges; and the auxiliary columns no
    longer hold a filter."""
    pg, cfg = product_cfg
    L = oracle.lib()
    assert L.orc_ctl_n_aux(1, 2431) == 4 and L.orc_ctl_n_aux(3, 45) == 2      # h_0 h_1 z_0 z_1 / z_0 z_1
    msgs = [b"abc", bytes(range(200))]
    rows, perms = sponge_and_keccak_work(oracle, msgs)
    rows_fewer, _ = sponge_and_keccak_work(oracle, msgs[:1])                   # the sponge table asks for one permutation only
    ir = real_ir({3, 4})
    a = o_state.txn_tables(ir, witness={3: perms, 4: rows})
    b = o_state.txn_tables(ir, witness={3: perms, 4: rows_fewer})              # same Keccak-f permutations, a smaller exposed subset
    cap = 4 << SMALL["stark_cap_height"]
    first = table_slices(a)[3][3]
    trace_cap = lambda tp: tp[first + 16:first + 16 + cap]
    assert not (trace_cap(a) == trace_cap(b)).all()                            # the subset is in the trace commitment ...
    assert not (a[2 + 13:2 + 13 + 4] == b[2 + 13:2 + 13 + 4]).all()            # ... that the lookup challenges are drawn from
    for tp in (a, b):
        assert o_state.verify_tables(tp) == 0
        pg.verify_txn_table_proofs(cfg, tp.tobytes(), gen_inputs=np.array(ir, dtype=np.uint64).tobytes())


def test_words_the_byte_packing_table_moves_are_memory_operations(oracle, o_state, product_cfg):
    """byte_packing -> memory with tables given by the caller: the packing rows and the memory log of the same strings
    (block_driver.memory_and_byte_packing_work_of_preimages) agree; a sequence that spells another word, or names an
    operation the log does not hold, leaves both tables valid alone and is rejected by both verifiers."""
    from proof_protocol_decoder_amd.block_driver import memory_and_byte_packing_work_of_preimages
    pg, cfg = product_cfg
    log, seqs = memory_and_byte_packing_work_of_preimages([b"hello, memory", bytes(range(70)), b"x" * 32])
    assert len(seqs) == 5 and len(log) == 10
    ir = real_ir({1, 6})
    good = o_state.txn_tables(ir, witness={1: seqs, 6: log})
    assert o_state.verify_tables(good) == 0
    pg.verify_txn_table_proofs(cfg, good.tobytes())
    packing, memory = first_row_openings(oracle, good, 1), first_row_openings(oracle, good, 6)
    assert (packing[0] == memory[0]).all() and tuple(packing[0]) != (1, 0)

    def rejected(bad_seqs, bad_log):
        with pytest.raises(RuntimeError, match="-12"):      # the prover refuses to go on
            o_state.txn_tables(ir, witness={1: bad_seqs, 6: bad_log})
        oracle.lib().orc_pg_set_prover_lookup_check(0)
        try:
            bad = o_state.txn_tables(ir, witness={1: bad_seqs, 6: bad_log})
        finally:
            oracle.lib().orc_pg_set_prover_lookup_check(1)
        assert o_state.verify_tables(bad) == -12
        with pytest.raises(pg.ProofGenError, match="cross-table lookup byte_packing -> memory does not hold") as e:
            pg.verify_txn_table_proofs(cfg, bad.tobytes())
        assert e.value.code == -5
    # one byte of one sequence differs: it still spells a word, the log is still a memory -- of another word
    s2 = [list(x) for x in seqs]
    s2[1][2] ^= 0x40
    rejected(s2, log)
    # the sequence names a timestamp at which the log holds no operation
    s3 = [list(x) for x in seqs]
    s3[2][0] += 5 << 8
    rejected(s3, log)
    # the log's read happens at another address than the sequence says (still sorted, still a memory)
    l4 = [list(x) for x in log] + [[0, 9, 1] + [7] * 8, [1, 9, 50] + [7] * 8]
    s4 = [list(x) for x in seqs]
    s4[4][1] = (s4[4][1] & 0xFF) | (9 << 8)
    rejected(s4, l4)


def test_table_proof_containers_are_bound_to_their_transcript(oracle, o_state, product_cfg):
    pg, cfg = product_cfg
    tp = o_state.txn_tables(real_ir({3, 4}))
    rng = np.random.default_rng(11)
    spots = [2 + 3, 2 + 13 + 1]                      # a public value, a lookup challenge
    for air, _, _, first, pw in table_slices(tp):    # and words of every table proof
        spots += [first + 16 + 1, first + int(rng.integers(16, pw))]
    for i in spots:
        bad = tp.copy()
        bad[i] ^= np.uint64(1 << int(rng.integers(0, 60)))
        assert o_state.verify_tables(bad) != 0
        with pytest.raises(pg.ProofGenError):
            pg.verify_txn_table_proofs(cfg, bad.tobytes())
    # the looked product moved to another value together with its looking partner is still refused (the running products
    # are tied to the committed columns by the table proofs)
    air, log_n, n_cols, first, pw = table_slices(tp)[4]
    with pytest.raises(pg.ProofGenError):
        pg.verify_txn_table_proofs(cfg, tp[:-1].tobytes())


def test_a_relabelled_table_is_refused_once_the_verifier_fixes_the_statement(oracle, o_state, product_cfg):
    """bp_verify_txn_table_proofs reads air_id / log_n / n_cols of every table from the blob -- the prover's
    word.  A prover that proves the Keccak-f table of a transaction as a SYNTHETIC table drops that table's constraints and
    the lookup that ties it to the sponge table, and the blob still verifies.  bp_verify_txn_table_proofs_for takes the
    statement from the transaction's IR (upstream's verifier owns all_stark) and refuses it; so is a blob whose public
    values are another transaction's."""
    pg, cfg = product_cfg
    irb = lambda words: np.array(words, dtype=np.uint64).tobytes()
    wanted = real_ir({3, 4})                      # the statement: Keccak-f and sponge tables proven with their AIRs
    honest = o_state.txn_tables(wanted)
    pg.verify_txn_table_proofs(cfg, honest.tobytes(), gen_inputs=irb(wanted))
    cheat_ir = real_ir({4})                       # the same transaction with table 3 "proven" as a synthetic table ...
    cheat_ir[18 + 3] = 2431                       # ... of the Keccak-f table's own width
    cheat = o_state.txn_tables(cheat_ir)
    pg.verify_txn_table_proofs(cfg, cheat.tobytes())          # the header is taken at its word: accepted
    with pytest.raises(pg.ProofGenError, match="table keccak is proven as AIR 0") as e:
        pg.verify_txn_table_proofs(cfg, cheat.tobytes(), gen_inputs=irb(wanted))
    assert e.value.code == -5
    other = real_ir({3, 4})
    other[3] += 1                                 # another transaction number: other public values
    with pytest.raises(pg.ProofGenError, match="public values"):
        pg.verify_txn_table_proofs(cfg, honest.tobytes(), gen_inputs=irb(other))
    shorter = real_ir({3, 4}, log_n=tuple(x + (1 if t == 3 else 0) for t, x in enumerate(LOG_N)))
    with pytest.raises(pg.ProofGenError):
        pg.verify_txn_table_proofs(cfg, honest.tobytes(), gen_inputs=irb(shorter))
