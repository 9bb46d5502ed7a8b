"""Block-level driver: what the reference leaves to an external scheduler ("Paladin",
docs/usage_seq_diagrams.md:8-20) -- fan the txns of a block out, aggregate, make the block proof.

One process per GPU.  Transactions are independent (trace_protocol.rs:18-22), so rank r proves the
CONTIGUOUS slice [r*n/N, (r+1)*n/N) (aggregation needs contiguous ranges, proof_types.rs:23-24) and
folds it into one proof with a local aggregation tree (aggregation_plan): no communication.  The only exchange step of
the whole path is the gather of the N sub-block proofs (a few hundred KB each) to rank 0, which
finishes the tree (N-1 aggregation proofs) and makes the block proof.  No field data ever crosses
GPUs, so there is no all-reduce here by construction.
"""
from concurrent.futures import ThreadPoolExecutor

from . import proof_gen as pg


def shard_bounds(n_items, rank, world_size):
    """Contiguous, balanced split: the first (n_items % world_size) ranks get one extra item."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


TREE_SHAPES = {"balanced": 0, "pairs_then_chain": 1}


def aggregation_plan(n, shape="balanced"):
    """The aggregation tree over n contiguous leaves (nodes 0 .. n-1): a list of (left, right) node ids, entry k being
    node n + k; the last entry is the root (bp_aggregation_plan, csrc/gi.cpp).  Aggregation needs contiguous ranges
    (proof_types.rs:23-24) and nothing else, the reference leaves the order to its scheduler
    (docs/usage_seq_diagrams.md:8-20).

    "balanced" (the default): adjacent pairs level by level, an odd tail carried up.
    "pairs_then_chain": adjacent leaves are paired and the pair results folded left to right, ((p0 p1) p2) p3 ...
    After the LAST leaf only its pair, one chain step and the block proof remain instead of log2(n) levels -- but the
    leaves of a shard do not finish one by one: the last wave of transactions (one per prover stream) ends together,
    and the chain then serialises what the balanced tree does in log2(streams) concurrent levels.  Measured on the
    32-txn shard: 36.9 against 37.3 txn-proofs/s for the balanced tree, level on the 256-txn block; kept as an option
    (bench.py --tree-shape)."""
    if n < 1:
        raise ValueError("nothing to aggregate")
    if shape not in TREE_SHAPES:
        raise ValueError("unknown tree shape %r" % (shape,))
    import ctypes as C
    L = pg._bind()
    L.bp_aggregation_plan.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.bp_aggregation_plan.restype = C.c_uint32
    pairs = (C.c_uint32 * max(2 * (n - 1), 1))()
    k = L.bp_aggregation_plan(n, TREE_SHAPES[shape], pairs)
    if k != n - 1:
        raise ValueError("bp_aggregation_plan refused n = %d, shape %r" % (n, shape))
    return [(pairs[2 * i], pairs[2 * i + 1]) for i in range(k)]


def run_shard(n, n_threads, shape, leaf_fn, agg_fn):
    """bp_run_shard (csrc/gi.cpp) over Python callables: the library's scheduler decides what runs when -- every
    aggregation the moment both of its children exist, AHEAD of the leaves still waiting for a thread -- and the
    callables do the work (leaf_fn(i) -> proof, agg_fn(lhs, rhs) -> proof; any Python objects: the buffers that travel
    through the scheduler carry node ids).  Returns (root, [leaves]).  The first exception raised by a callable ends
    the run and is re-raised."""
    import ctypes as C
    import itertools
    import threading
    L = pg._bind()
    libc = C.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes = C.c_void_p, [C.c_size_t]
    u8pp, szp = C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)
    LEAF = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, u8pp, szp)
    AGG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.c_int, C.POINTER(C.c_uint8), C.c_size_t, C.c_int, u8pp, szp)
    results, errors, lock, ids = {}, [], threading.Lock(), itertools.count(n)

    def emit(nid, value, out, out_len):
        with lock:
            results[nid] = value
        buf = libc.malloc(8)
        C.memmove(buf, nid.to_bytes(8, "little"), 8)
        out[0] = C.cast(buf, C.POINTER(C.c_uint8))
        out_len[0] = 8

    def guarded(f):
        def g(*a):
            try:
                return f(*a)
            except BaseException as e:   # surface the first failure to the caller
                with lock:
                    errors.append(e)
                return -4
        return g

    @guarded
    def leaf(_ctx, i, out, out_len):
        emit(i, leaf_fn(i), out, out_len)
        return 0

    @guarded
    def agg(_ctx, l, _ln, _la, r, _rn, _ra, out, out_len):
        li, ri = int.from_bytes(C.string_at(l, 8), "little"), int.from_bytes(C.string_at(r, 8), "little")
        with lock:
            a, b = results[li], results[ri]
        emit(next(ids), agg_fn(a, b), out, out_len)
        return 0

    class Opt(C.Structure):
        _fields_ = [("n_threads", C.c_uint32), ("tree_shape", C.c_uint32)]
    L.bp_run_shard.argtypes = [C.c_uint32, C.POINTER(Opt), LEAF, AGG, C.c_void_p, C.c_void_p, u8pp, szp, C.c_void_p, C.c_void_p]
    root, root_len = C.POINTER(C.c_uint8)(), C.c_size_t()
    opt = Opt(max(1, n_threads), TREE_SHAPES[shape])
    rc = L.bp_run_shard(n, C.byref(opt), LEAF(leaf), AGG(agg), None, None, C.byref(root), C.byref(root_len), None, None)
    if errors:
        raise errors[0]
    pg.check(rc)
    top = results[int.from_bytes(pg.take_buffer(root, root_len), "little")]
    return top, [results[i] for i in range(n)]


def tree_reduce(proofs, agg_fn, pool=None, shape="balanced"):
    """Fold a contiguous list of aggregatable proofs into one along aggregation_plan(len(proofs), shape).
    Aggregations whose children exist are independent and run concurrently on `pool`."""
    nodes = list(proofs)
    if not nodes:
        raise ValueError("nothing to aggregate")
    plan = aggregation_plan(len(nodes), shape)
    n = len(nodes)
    nodes += [None] * len(plan)
    todo = list(range(len(plan)))
    while todo:
        ready = [k for k in todo if nodes[plan[k][0]] is not None and nodes[plan[k][1]] is not None]
        if pool is not None and len(ready) > 1:
            outs = list(pool.map(lambda k: agg_fn(nodes[plan[k][0]], nodes[plan[k][1]]), ready))
        else:
            outs = [agg_fn(nodes[plan[k][0]], nodes[plan[k][1]]) for k in ready]
        for k, o in zip(ready, outs):
            nodes[n + k] = o
        todo = [k for k in todo if k not in set(ready)]
    return nodes[-1]


class TorchGather:
    """Gather variable-length proof bytes to rank 0 with torch.distributed (backend nccl = RCCL over
    xGMI on the GPU box, gloo in the CPU tests).  Lengths first, then one padded uint8 gather."""

    def __init__(self, device):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.device = torch, dist, device

    def gather_bytes(self, payload, failed=False):
        """payload may be empty (a rank without transactions: rank 0 gets b"" for it).  `failed` marks a rank
        whose shard raised: the length exchange doubles as the status exchange (length -1), and then EVERY
        rank raises ShardFailed before the data gather, so nobody is left waiting in a collective."""
        torch, dist = self.torch, self.dist
        world, rank = dist.get_world_size(), dist.get_rank()
        n = torch.tensor([-1 if failed else len(payload)], dtype=torch.int64, device=self.device)
        lens = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(lens, n)
        lens = [int(x.item()) for x in lens]
        bad = [r for r, l in enumerate(lens) if l < 0]
        if bad:
            raise ShardFailed("shard proving failed on rank(s) %s" % bad)
        max_len = max(max(lens), 1)
        buf = torch.zeros(max_len, dtype=torch.uint8, device=self.device)
        if payload:
            buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(self.device)
        out = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, out, dst=0)
        if rank != 0:
            return None
        return [bytes(out[r][:lens[r]].cpu().numpy().tobytes()) for r in range(world)]


    # ---- the pieces of the pairwise top tree (prove_block_distributed, top_tree="pairwise")
    def exchange_status(self, n_bytes, failed=False):
        """all_gather of every rank's payload length (-1 = its shard failed): every rank raises ShardFailed when any
        failed, before anybody waits for a payload that will never come.  Returns the lengths."""
        torch, dist = self.torch, self.dist
        n = torch.tensor([-1 if failed else n_bytes], dtype=torch.int64, device=self.device)
        lens = [torch.zeros_like(n) for _ in range(dist.get_world_size())]
        dist.all_gather(lens, n)
        lens = [int(x.item()) for x in lens]
        bad = [r for r, l in enumerate(lens) if l < 0]
        if bad:
            raise ShardFailed("shard proving failed on rank(s) %s" % bad)
        return lens

    def send_bytes(self, dst, payload, failed=False):
        """length (-1 = this rank failed inside the tree), then the bytes"""
        torch, dist = self.torch, self.dist
        dist.send(torch.tensor([-1 if failed else len(payload)], dtype=torch.int64, device=self.device), dst)
        if not failed and payload:
            dist.send(torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(self.device), dst)

    def recv_bytes(self, src):
        torch, dist = self.torch, self.dist
        n = torch.zeros(1, dtype=torch.int64, device=self.device)
        dist.recv(n, src)
        n = int(n.item())
        if n < 0:
            raise ShardFailed("aggregation failed on rank %d" % src)
        if n == 0:
            return b""
        buf = torch.empty(n, dtype=torch.uint8, device=self.device)
        dist.recv(buf, src)
        return bytes(buf.cpu().numpy().tobytes())


class ShardFailed(RuntimeError):
    """Raised on every rank when any rank's shard failed (see TorchGather.gather_bytes)."""


class BlockDriver:
    """prove_txn(ir) -> GeneratedTxnProof, prove_agg(lhs, rhs) -> GeneratedAggProof,
    prove_block(parent_or_None, agg) -> GeneratedBlockProof.  The defaults bind the HIP library."""

    def __init__(self, p_state=None, n_threads=4, prove_txn=None, prove_agg=None, prove_block=None,
                 decode_proof=None, tree_shape="balanced"):
        self.n_threads = n_threads
        self.tree_shape = tree_shape   # of a shard's local tree (aggregation_plan)
        self.pool = ThreadPoolExecutor(n_threads) if n_threads > 1 else None
        # the HIP prover state when no callable overrides it: prove_shard then runs inside the library (bp_prove_shard)
        self._native = p_state if (p_state is not None and prove_txn is None and prove_agg is None) else None
        self.prove_txn = prove_txn or (lambda ir: pg.generate_txn_proof(p_state, ir))
        self.prove_agg = prove_agg or (lambda a, b: pg.generate_agg_proof(p_state, a, b))
        self.prove_block = prove_block or (lambda parent, agg: pg.generate_block_proof(p_state, parent, agg))
        self.decode_proof = decode_proof or self._decode

    @staticmethod
    def _decode(raw):
        pv, kind = pg.public_values_of(raw)
        return (pg.GeneratedAggProof if kind == 1 else pg.GeneratedTxnProof)(pv, raw)

    def prove_shard(self, irs, shape=None):
        """All txn proofs of a contiguous slice and its local aggregation tree (aggregation_plan).  Every aggregation
        starts the moment both of its children exist and goes AHEAD of the transactions still waiting for a thread,
        so the tree advances with the proving instead of piling up behind it.  The policy is the library's
        (csrc/gi.cpp): bp_prove_shard when this driver binds the HIP prover and the IRs carry no witness data of their
        own, else bp_run_shard over this driver's callables."""
        n = len(irs)
        shape = shape or self.tree_shape
        if n < 1:
            raise ValueError("nothing to aggregate")
        if self._native is not None and all(isinstance(ir, (bytes, pg.TxnProofGenIR)) and getattr(ir, "witness", None) is None
                                            and getattr(ir, "keccak_inputs", None) is None for ir in irs):
            return self._prove_shard_native(irs, shape)
        return run_shard(n, self.n_threads if n > 1 else 1, shape, lambda i: self.prove_txn(irs[i]), self.prove_agg)

    def _prove_shard_native(self, irs, shape):
        import ctypes as C
        L = pg._bind()
        raw = b"".join(ir.to_bytes() if isinstance(ir, pg.TxnProofGenIR) else bytes(ir) for ir in irs)
        n = len(irs)

        class Opt(C.Structure):
            _fields_ = [("n_threads", C.c_uint32), ("tree_shape", C.c_uint32)]
        u8p = C.POINTER(C.c_uint8)
        L.bp_prove_shard.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.POINTER(Opt), C.c_void_p,
                                     C.POINTER(u8p), C.POINTER(C.c_size_t), C.POINTER(u8p), C.POINTER(C.c_size_t)]
        root, root_len = u8p(), C.c_size_t()
        leaves, lens = (u8p * n)(), (C.c_size_t * n)()
        opt = Opt(self.n_threads, TREE_SHAPES[shape])
        pg.check(L.bp_prove_shard(self._native._h, raw, len(raw) // n, n, C.byref(opt), None, C.byref(root), C.byref(root_len),
                                  leaves, lens))
        txns = [self._decode(pg.take_buffer(leaves[i], C.c_size_t(lens[i]))) for i in range(n)]
        return self._decode(pg.take_buffer(root, root_len)), txns

    def _aggregate_native(self, subs):
        """the top of the block's tree over proofs made elsewhere (bp_aggregate_proofs)"""
        import ctypes as C
        if len(subs) == 1:
            return subs[0]
        L = pg._bind()

        class Opt(C.Structure):
            _fields_ = [("n_threads", C.c_uint32), ("tree_shape", C.c_uint32)]
        n = len(subs)
        raws = [bytes(p.intern) for p in subs]
        ptrs = (C.c_char_p * n)(*raws)
        lens = (C.c_size_t * n)(*[len(r) for r in raws])
        L.bp_aggregate_proofs.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_uint32, C.POINTER(Opt),
                                          C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        out, out_len = C.POINTER(C.c_uint8)(), C.c_size_t()
        opt = Opt(self.n_threads, 0)
        pg.check(L.bp_aggregate_proofs(self._native._h, ptrs, lens, n, C.byref(opt), C.byref(out), C.byref(out_len)))
        return self._decode(pg.take_buffer(out, out_len))

    def prove_shard_gi(self, geni, first, n, gi_options, shape=None):
        """The same for entries [first, first + n) of a decoded block ("BPGGENI1" bytes, decoding.into_txn_proof_gen_ir
        raw form): bp_prove_shard_gi derives every entry's IR and witness in the library (csrc/gi.cpp).
        gi_options: a GiOptions."""
        import ctypes as C
        L = pg._bind()

        class Opt(C.Structure):
            _fields_ = [("n_threads", C.c_uint32), ("tree_shape", C.c_uint32)]
        u8p = C.POINTER(C.c_uint8)
        L.bp_prove_shard_gi.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(GiOptions),
                                        C.POINTER(Opt), C.c_void_p, C.POINTER(u8p), C.POINTER(C.c_size_t), C.POINTER(u8p),
                                        C.POINTER(C.c_size_t)]
        root, root_len = u8p(), C.c_size_t()
        leaves, lens = (u8p * max(n, 1))(), (C.c_size_t * max(n, 1))()
        opt = Opt(self.n_threads, TREE_SHAPES[shape or self.tree_shape])
        pg.check(L.bp_prove_shard_gi(self._native._h, geni, len(geni), first, n, C.byref(gi_options), C.byref(opt), None,
                                     C.byref(root), C.byref(root_len), leaves, lens))
        txns = [self._decode(pg.take_buffer(leaves[i], C.c_size_t(lens[i]))) for i in range(n)]
        return self._decode(pg.take_buffer(root, root_len)), txns

    def prove_block_distributed(self, irs, rank=0, world_size=1, gather=None, parent=None, top_tree="pairwise"):
        """Returns the GeneratedBlockProof on rank 0, None elsewhere.

        A block with fewer entries than ranks (decoding.rs:304-347 pads to two, so any block of 0 or 1
        transactions on an 8-GPU node) leaves the high ranks without a slice: they take part with an empty payload
        that is skipped.  A rank whose shard raises reports it through the length exchange, and every rank raises
        instead of waiting for a payload that will never come.

        top_tree: how the N sub-block proofs become one (the same balanced tree over ranks either way, so the same bytes):
          "pairwise" (default): level by level, rank 2k+1 sends its proof to rank 2k, which aggregates (then 4k+2 -> 4k,
            ...): every rank verifies ONE foreign child per level and rank 0 makes log2 N aggregation proofs one after
            the other;
          "gather": all N proofs to rank 0, which makes the N - 1 aggregations alone (rounds 1-4)."""
        lo, hi = shard_bounds(len(irs), rank, world_size)
        sub, err = None, None
        try:
            if hi > lo:
                sub, _ = self.prove_shard(irs[lo:hi])
        except Exception as e:  # held until the other ranks know
            err = e
        if world_size > 1 and top_tree == "pairwise":
            top = self._pairwise_top_tree(sub, err, rank, world_size, gather)
            if rank != 0:
                return None
        elif world_size > 1:
            try:
                raws = gather.gather_bytes(sub.intern if sub is not None else b"", failed=err is not None)
            except ShardFailed:
                if err is not None:
                    raise err
                raise
            if rank != 0:
                return None
            subs = [self.decode_proof(r) for r in raws if r]
            if not subs:
                raise ValueError("a block needs at least two transactions (decoding.rs:304-347 pads to >= 2)")
            top = self._aggregate_native(subs) if self._native is not None else tree_reduce(subs, self.prove_agg, self.pool)
        else:
            if err is not None:
                raise err
            top = sub
        if not isinstance(top, pg.GeneratedAggProof):
            raise ValueError("a block needs at least two transactions (decoding.rs:304-347 pads to >= 2)")
        return self.prove_block(parent, top)

    def _pairwise_top_tree(self, sub, err, rank, world_size, gather):
        """Level l: rank r with r % 2^(l+1) == 2^l sends what it holds to rank r - 2^l and is done; the receiver
        aggregates (its own range is the left child: rank order is transaction order).  The balanced tree over ranks,
        odd tails carried up -- aggregation_plan(N)'s shape.  A failure inside the tree travels up as a length of -1, so
        no parent waits for a payload that will never come; the failing rank raises its own error, the ranks above it
        ShardFailed."""
        try:
            gather.exchange_status(len(sub.intern) if sub is not None else 0, failed=err is not None)
        except ShardFailed:
            if err is not None:
                raise err
            raise
        cur, step = sub, 1
        while step < world_size:
            if rank % (2 * step) == step:
                gather.send_bytes(rank - step, cur.intern if (cur is not None and err is None) else b"", failed=err is not None)
                if err is not None:
                    raise err
                return None
            src = rank + step
            if src < world_size:
                try:
                    raw = gather.recv_bytes(src)
                    if err is None and raw:
                        other = self.decode_proof(raw)
                        cur = other if cur is None else self.prove_agg(cur, other)
                except Exception as e:   # keep walking the levels: the ranks above must hear of it
                    err = err or e
            step *= 2
        if err is not None:
            raise err
        return cur

    def close(self):
        if self.pool is not None:
            self.pool.shutdown()


def synthetic_block_irs(block_number, n_txns, table_log_n, table_width, seed_base=0x5EED000000000000,
                        root0=(1, 2, 3, 4), keccak_air=False, logic_air=False, memory_air=False, arithmetic_air=False,
                        byte_packing_air=False, keccak_sponge_air=False, arithmetic_mul_air=False):
    """The synthetic block of SURVEY.md section 8(d): n_txns txns with distinct seeds whose public
    values chain (state root, txn number, gas) like decoding.rs:106-154 chains GenerationInputs.
    keccak_air: every transaction's Keccak table (index 3) is a real Keccak-f[1600] trace (AIR 1; the table's width
    becomes 2431).  logic_air / memory_air: likewise the logic table (index 5) with the logic AIR (AIR 2; width 524) and
    the memory table (index 6) with the memory AIR (AIR 3; width 45); arithmetic_air: the arithmetic table (index 0)
    with the arithmetic AIR (AIR 4; width 309); byte_packing_air: the byte-packing table (index 1) with AIR 5 (width 299);
    keccak_sponge_air: the Keccak sponge table (index 4) with AIR 6 (width 2414); arithmetic_mul_air (instead of
    arithmetic_air): the arithmetic table with the multiplication AIR (AIR 7; width 1217)."""
    if keccak_air:
        table_width = tuple(2431 if t == 3 else w for t, w in enumerate(table_width))
    if arithmetic_mul_air:
        table_width = tuple(1217 if t == 0 else w for t, w in enumerate(table_width))
    if logic_air:
        table_width = tuple(524 if t == 5 else w for t, w in enumerate(table_width))
    if memory_air:
        table_width = tuple(45 if t == 6 else w for t, w in enumerate(table_width))
    if arithmetic_air:
        table_width = tuple(309 if t == 0 else w for t, w in enumerate(table_width))
    if byte_packing_air:
        table_width = tuple(299 if t == 1 else w for t, w in enumerate(table_width))
    if keccak_sponge_air:
        table_width = tuple(2414 if t == 4 else w for t, w in enumerate(table_width))
    import ctypes as C
    L = pg._bind()
    L.bp_state_root_after.argtypes = [C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    irs, root, gas = [], tuple(root0), 0
    for i in range(n_txns):
        seed = seed_base + (block_number << 20) + i
        irs.append(pg.TxnProofGenIR(block_number, i, gas, gas + 21000, root, seed, tuple(table_log_n),
                                    tuple(table_width), keccak_air=keccak_air, logic_air=logic_air,
                                    memory_air=memory_air, arithmetic_air=arithmetic_air,
                                    byte_packing_air=byte_packing_air, keccak_sponge_air=keccak_sponge_air,
                                    arithmetic_mul_air=arithmetic_mul_air))
        out = (C.c_uint64 * 4)()
        pg.check(L.bp_state_root_after((C.c_uint64 * 4)(*root), seed, i, out))
        root, gas = tuple(out), gas + 21000
    return irs


DUMMY_SEED = 0x44554D4D59000000   # "DUMMY": witness seed of padding entries


def pad_with_dummy_irs(irs, block_number, state_root, table_log_n, table_width, has_withdrawals=False):
    """`pad_gen_inputs_with_dummy_inputs_if_needed` (protocol_decoder/src/decoding.rs:304-347): an aggregation
    needs two entries and a block proof needs an aggregation, so a block of 0 transactions gets two dummy entries
    and a block of 1 gets one -- BEFORE the transaction, or AFTER it when the block has withdrawals (the
    withdrawals then ride on that last dummy, :356-402).  Dummy entries do not change state (:484-520).
    `state_root` is the block's initial state root (used when there is no transaction to take it from).
    Returns (irs, dummies_added).

    One deliberate difference: the reference builds the prepended dummy from the counters AFTER the real
    transaction (`extra_data`, :329-333: txn_number_before = 1, gas = the block's total); here it carries the
    counters of its position (0, 0) so that txn number, gas and state root chain from entry to entry, which
    `generate_agg_proof` checks (proof_types.rs:23-24).
    """
    irs = list(irs)
    mk = lambda txn_no, gas, root, k: pg.TxnProofGenIR(block_number, txn_no, gas, gas, tuple(root), DUMMY_SEED + k,
                                                       tuple(table_log_n), tuple(table_width), dummy=True)
    if len(irs) == 0:
        return [mk(0, 0, state_root, 0), mk(0, 0, state_root, 1)], True
    if len(irs) == 1:
        only = irs[0]
        if has_withdrawals:   # dummy after the only txn: starts where the txn ended
            after = pg.state_root_after(only.state_root_before, only.seed, only.txn_number_before)
            return [only, mk(only.txn_number_before + 1, only.gas_used_after, after, 0)], True
        return [mk(only.txn_number_before, only.gas_used_before, only.state_root_before, 0), only], True
    return irs, False


def irs_from_block_trace(block_trace, block_number, table_log_n, table_width, has_withdrawals=False):
    """Front door: a `trace_protocol.BlockTrace` payload -> the synthetic IRs of its transactions.

    Mirrors the sequencing of `ProcessedBlockTrace::into_txn_proof_gen_ir` (protocol_decoder/src/decoding.rs:81-177):
    transactions in payload order, `txn_number_before` = index, `gas_used_before/after` accumulated from
    `TxnMeta.gas_used` (:122-124, :150-152), the state root threaded from one txn to the next (:129), then the
    dummy padding of `pad_with_dummy_irs` (:157-164).  What the reference derives with its MPT machinery is
    derived synthetically here: the chain starts at the state root of the decoded compact witness
    (folded into four field elements) and each txn's witness seed is a hash of its payload bytes.  Withdrawals
    only steer where the padding goes (`has_withdrawals`); their balance updates (:404-428) are not modelled.
    """
    from . import compact
    pre = block_trace.process_pre_images()
    P = 0xFFFFFFFF00000001
    root0 = root = tuple(int.from_bytes(pre.state_root[8 * i:8 * i + 8], "little") % P for i in range(4))
    irs, gas = [], 0
    for i, info in enumerate(block_trace.txn_info):
        m = info.meta
        digest = compact.keccak256(m.byte_code + m.new_txn_trie_node_byte + m.new_receipt_trie_node_byte)
        seed = int.from_bytes(digest[:8], "little")
        irs.append(pg.TxnProofGenIR(block_number, i, gas, gas + m.gas_used, root, seed, tuple(table_log_n),
                                    tuple(table_width)))
        root, gas = pg.state_root_after(root, seed, i), gas + m.gas_used
    return pad_with_dummy_irs(irs, block_number, root0, table_log_n, table_width, has_withdrawals)[0]


def keccak_inputs_of_generation_inputs(g, trie_nodes=False):
    """The Keccak-f permutations a transaction's own data asks for: Keccak-256 of `signed_txn` (the transaction hash)
    and of every `contract_code` entry (the code hashes the decoder keys them by, decoding.rs:131-145), as the states
    that go into the permutations, in that order (code in ascending hash order).
    trie_nodes: then also the hashing of the entry's partial tries (decoding.rs:179-217: state, transactions, receipts,
    storage tries in their order) -- every node that is referenced by hash, children before parents, which is the bulk
    of what a zkEVM's Keccak table holds; each trie's last digest is checked against its root.  What remains
    upstream-only is the hashing done WHILE executing (KECCAK256 opcodes, the tries after the transaction)."""
    from .partial_trie import hashed_node_preimages
    states = []
    if g.signed_txn:
        states += pg.keccak256_permutation_inputs(bytes(g.signed_txn))[1]
    for h in sorted(g.contract_code):
        digest, st = pg.keccak256_permutation_inputs(bytes(g.contract_code[h]))
        if digest != bytes(h):
            raise ValueError("contract_code is keyed by a hash that is not the Keccak-256 of its bytes")
        states += st
    if trie_nodes:
        tries = [g.tries.state_trie, g.tries.transactions_trie, g.tries.receipts_trie] + [t for _, t in g.tries.storage_tries]
        for trie in tries:
            digest = None
            for enc in hashed_node_preimages(trie.root):
                digest, st = pg.keccak256_permutation_inputs(enc)
                states += st
            if digest is not None and digest != trie.hash():
                raise ValueError("a partial trie's node hashes do not end in its root")
    return states


def hashed_preimages_of_generation_inputs(g, trie_nodes=False):
    """The byte strings keccak_inputs_of_generation_inputs hashes, in its order: signed_txn, the contract codes in
    ascending hash order and, with trie_nodes, the hash-referenced nodes of the entry's partial tries."""
    from .partial_trie import hashed_node_preimages
    out = [bytes(g.signed_txn)] if g.signed_txn else []
    out += [bytes(g.contract_code[h]) for h in sorted(g.contract_code)]
    if trie_nodes:
        for trie in [g.tries.state_trie, g.tries.transactions_trie, g.tries.receipts_trie] + [t for _, t in g.tries.storage_tries]:
            out += hashed_node_preimages(trie.root)
    return out


def memory_and_byte_packing_work_of_preimages(preimages):
    """What moving those byte strings to the hasher looks like in a zkEVM, as witness data for the byte-packing table
    (AIR 5) and the memory table (AIR 3), tied by the lookup byte_packing -> memory (AIRS.md section 3): the hasher
    takes the bytes 32 at a time -- one byte-packing sequence per chunk (the last chunk of a string may be shorter):
    [is_read = 1 | timestamp << 8, len | address << 8, four words of byte slots] -- and the 256-bit word a chunk spells
    lives at its own address (the chunks numbered through the strings), written once and read once by the packer: the
    memory log [is_read, address, timestamp, eight 32-bit value limbs], sorted by (address, timestamp).  The read is the
    operation the packing row names."""
    log, seqs, addr = [], [], 0
    for m in preimages:
        for off in range(0, len(m), 32):
            chunk = m[off:off + 32]
            word = int.from_bytes(chunk, "big")
            limbs = [(word >> (32 * k)) & 0xFFFFFFFF for k in range(8)]
            ts = addr + 2
            log.append([0, addr, 1] + limbs)       # the word is stored ...
            log.append([1, addr, ts] + limbs)      # ... and read back by the packer
            padded = chunk + bytes(32 - len(chunk))
            seqs.append([1 | (ts << 8), len(chunk) | (addr << 8)]
                        + [int.from_bytes(padded[8 * w:8 * w + 8], "little") for w in range(4)])
            addr += 1
    return log, seqs


import ctypes as _C


class GiOptions(_C.Structure):
    """bp_gi_options (include/bpg.h)"""
    _fields_ = [("block_number", _C.c_uint64), ("table_log_n", _C.c_uint32 * 7), ("table_width", _C.c_uint32 * 7),
                ("flags", _C.c_uint32)]
    KECCAK_AIR, KECCAK_TRIE_NODES, MEMORY_AIR, BYTE_PACKING_AIR, KECCAK_SPONGE_AIR, LOGIC_AIR = 1, 2, 4, 8, 16, 32

    @staticmethod
    def make(block_number, table_log_n, table_width, keccak_air=False, keccak_trie_nodes=False, memory_air=False,
             byte_packing_air=False, keccak_sponge_air=False, logic_air=False):
        return GiOptions(block_number, (_C.c_uint32 * 7)(*table_log_n), (_C.c_uint32 * 7)(*table_width),
                         (1 if keccak_air else 0) | (2 if keccak_trie_nodes else 0) | (4 if memory_air else 0)
                         | (8 if byte_packing_air else 0) | (16 if keccak_sponge_air else 0) | (32 if logic_air else 0))


class GiChain(_C.Structure):
    """bp_gi_chain (include/bpg.h): txn number, gas and state root where an entry starts"""
    _fields_ = [("txn_number", _C.c_uint64), ("gas_used", _C.c_uint64), ("state_root", _C.c_uint64 * 4)]


def gi_irs(geni, gi_options):
    """The 25-word IRs of every entry of a decoded block as the library derives them (bp_gi_chain_start, bp_gi_entry_ir):
    what irs_from_generation_inputs computes in Python, as bytes."""
    L = pg._bind()
    n = _C.c_uint32()
    L.bp_gi_count.argtypes = [_C.c_char_p, _C.c_size_t, _C.POINTER(_C.c_uint32)]
    L.bp_gi_chain_start.argtypes = [_C.c_char_p, _C.c_size_t, _C.POINTER(GiChain)]
    L.bp_gi_entry_ir.argtypes = [_C.c_char_p, _C.c_size_t, _C.c_uint32, _C.POINTER(GiOptions), _C.POINTER(GiChain),
                                 _C.POINTER(_C.c_uint64)]
    pg.check(L.bp_gi_count(geni, len(geni), _C.byref(n)))
    chain = GiChain()
    pg.check(L.bp_gi_chain_start(geni, len(geni), _C.byref(chain)))
    out = []
    for k in range(n.value):
        ir = (_C.c_uint64 * 25)()
        pg.check(L.bp_gi_entry_ir(geni, len(geni), k, _C.byref(gi_options), _C.byref(chain), ir))
        out.append(bytes(ir))
    return out


def generate_txn_proof_gi(p_state, geni, entry, gi_options, chain, abort_signal=None):
    """generate_txn_proof(&ProverState, GenerationInputs, abort) (proof_gen.rs:39-43) for entry `entry` of a decoded
    block: bp_generate_txn_proof_gi.  chain: a GiChain holding what the entries before left (gi_chain_start for entry
    0); moved past the entry."""
    L = pg._bind()
    out, n = pg._out()
    L.bp_generate_txn_proof_gi.argtypes = [_C.c_void_p, _C.c_char_p, _C.c_size_t, _C.c_uint32, _C.POINTER(GiOptions),
                                           _C.POINTER(GiChain), _C.c_void_p, _C.POINTER(_C.POINTER(_C.c_uint8)),
                                           _C.POINTER(_C.c_size_t)]
    pg.check(L.bp_generate_txn_proof_gi(p_state._h, geni, len(geni), entry, _C.byref(gi_options), _C.byref(chain),
                                        _C.byref(abort_signal) if abort_signal is not None else None, _C.byref(out), _C.byref(n)))
    intern = pg.take_buffer(out, n)
    return pg.GeneratedTxnProof(pg.public_values_of(intern)[0], intern)


def gi_chain_start(geni):
    L = pg._bind()
    L.bp_gi_chain_start.argtypes = [_C.c_char_p, _C.c_size_t, _C.POINTER(GiChain)]
    chain = GiChain()
    pg.check(L.bp_gi_chain_start(geni, len(geni), _C.byref(chain)))
    return chain


def irs_from_generation_inputs(gen_inputs, block_number, table_log_n, table_width, keccak_air=False,
                               keccak_trie_nodes=False, memory_air=False, byte_packing_air=False, keccak_sponge_air=False,
                               logic_air=False):
    """`Vec<TxnProofGenIR>` as produced by `decoding.into_txn_proof_gen_ir` (the reference's
    BlockTrace::into_txn_proof_gen_ir: minimal tries, delta replay, dummy padding, withdrawals) -> the IRs this
    library's prover takes.  The zkEVM that would consume the partial tries is upstream-only, so
    each entry is bound to its proof through the witness seed: seed = keccak(signed_txn | state, transactions and
    receipts roots after | withdrawals), i.e. a different decoded state transition gives a different proof.  Txn
    number and gas come from the decoded entries; the state-root public value starts at the decoded pre-state
    root (folded into four field elements) and chains entry to entry.  Entries without a transaction (dummy
    padding, the withdrawal carrier) become dummy IRs: proven, counters do not advance (decoding.rs:484-520) --
    a prepended dummy is renumbered to its position, as pad_with_dummy_irs documents.
    keccak_air: every entry's Keccak table (index 3) becomes a real Keccak-f[1600] trace (AIR 1, 2431 columns) whose
    permutations are the entry's OWN hashing work (keccak_inputs_of_generation_inputs): the table then attests data of
    the decoded transaction, not only a seed; its height grows to hold them (24 rows per permutation).
    keccak_trie_nodes: the hashing of the entry's partial tries is part of that work (the prover state's Keccak range
    must then reach the taller tables).
    memory_air / byte_packing_air (with keccak_air): the memory table (AIR 3) and the byte-packing table (AIR 5) of
    every entry hold the traffic of the SAME bytes on their way to the hasher
    (memory_and_byte_packing_work_of_preimages) instead of a seeded witness; their heights grow to hold it.
    keccak_sponge_air (with keccak_air): the Keccak sponge table (AIR 6) absorbs the same strings block by block
    (pg.keccak256_sponge_rows): its (xored rate, capacity) -> updated state pairs are, row for row, the inputs and
    outputs of the Keccak table's permutations.
    logic_air (with keccak_sponge_air): the logic table (AIR 2) holds the sponge rows' XORs first -- five operations per
    row of the sponge table (the lookup keccak_sponge -> logic derives them from the sponge table's trace inside the
    library); its height grows to hold them."""
    from . import compact
    P = 0xFFFFFFFF00000001
    first = gen_inputs[0].tries.state_trie.hash()
    root = tuple(int.from_bytes(first[8 * i:8 * i + 8], "little") % P for i in range(4))
    irs, txn_no, gas = [], 0, 0
    if keccak_air:
        table_width = tuple(2431 if t == 3 else w for t, w in enumerate(table_width))
    if (memory_air or byte_packing_air or keccak_sponge_air) and not keccak_air:
        raise ValueError("the memory / byte-packing / sponge work is that of the hashed bytes: it needs keccak_air")
    if keccak_sponge_air:
        table_width = tuple(2414 if t == 4 else w for t, w in enumerate(table_width))
    if logic_air:
        if not keccak_sponge_air:
            raise ValueError("the logic table's work is the sponge table's XORs: logic_air needs keccak_sponge_air")
        table_width = tuple(524 if t == 5 else w for t, w in enumerate(table_width))
    if memory_air:
        table_width = tuple(45 if t == 6 else w for t, w in enumerate(table_width))
    if byte_packing_air:
        table_width = tuple(299 if t == 1 else w for t, w in enumerate(table_width))
    base_log_n = tuple(table_log_n)
    for k, g in enumerate(gen_inputs):
        kw = {}
        table_log_n = base_log_n
        if keccak_air:
            states = keccak_inputs_of_generation_inputs(g, trie_nodes=keccak_trie_nodes)
            need = max(24 * len(states), 1)
            table_log_n = tuple(max(l, (need - 1).bit_length()) if t == 3 else l for t, l in enumerate(base_log_n))
            kw = dict(keccak_air=True, keccak_inputs=tuple(tuple(s) for s in states))
            if memory_air or byte_packing_air or keccak_sponge_air:
                pre = hashed_preimages_of_generation_inputs(g, trie_nodes=keccak_trie_nodes)
                log, seqs = memory_and_byte_packing_work_of_preimages(pre)
                wit, ln = [], list(table_log_n)
                if keccak_sponge_air:
                    rows = [r for m in pre for r in pg.keccak256_sponge_rows(m)[1]]
                    ln[4] = max(ln[4], (max(len(rows), 1) - 1).bit_length())
                    wit.append((4, tuple(tuple(r) for r in rows)))
                    if logic_air:
                        ln[5] = max(ln[5], ((5 << ln[4]) - 1).bit_length())
                if memory_air:
                    ln[6] = max(ln[6], (max(len(log), 1) - 1).bit_length())
                    wit.append((6, tuple(tuple(r) for r in log)))
                if byte_packing_air:
                    ln[1] = max(ln[1], (max(len(seqs), 1) - 1).bit_length())
                    wit.append((1, tuple(tuple(r) for r in seqs)))
                table_log_n = tuple(ln)
                kw.update(memory_air=memory_air, byte_packing_air=byte_packing_air, keccak_sponge_air=keccak_sponge_air,
                          logic_air=logic_air, witness=tuple(wit))
        r = g.trie_roots_after
        blob = (g.signed_txn or b"") + r.state_root + r.transactions_root + r.receipts_root
        blob += b"".join(bytes(a) + int(v).to_bytes(32, "big") for a, v in g.withdrawals)
        seed = int.from_bytes(compact.keccak256(blob)[:8], "little")
        if g.signed_txn is None:
            irs.append(pg.TxnProofGenIR(block_number, txn_no, gas, gas, root, seed, tuple(table_log_n), tuple(table_width),
                                        dummy=True, **kw))
            continue
        used = g.gas_used_after - g.gas_used_before
        irs.append(pg.TxnProofGenIR(block_number, txn_no, gas, gas + used, root, seed, tuple(table_log_n),
                                    tuple(table_width), **kw))
        root, txn_no, gas = pg.state_root_after(root, seed, txn_no), txn_no + 1, gas + used
    return irs
