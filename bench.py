#!/usr/bin/env python3
"""bench.py -- txn-proofs/sec on the 256-txn synthetic block (BASELINE.json metric), one process
per GPU.

A step = one whole 256-txn block: every rank proves its contiguous slice of the block's txns and
its local aggregation subtree; the N sub-block proofs are gathered to rank 0 over RCCL, which
finishes the tree and makes the block proof (strong scaling: the block is fixed, the slice shrinks
with N).  value = txns * steps / wall time, inputs (the IRs) resident before the clock starts; the
witness of each txn is generated on the GPU inside generate_txn_proof, as generate_traces runs
inside the reference's call (proof_gen.rs:44-52).  `python3 bench.py --gpus N` without a launcher
starts its own N ranks (torch.distributed.run) before it touches the GPU.

The timed region runs first.  Everything that is measured ALONE on the chip runs afterwards in a
fresh child process (`--leg-only`) and is merged into the one JSON line:
  roofline      -- the coset-LDE NTT kernel family (the HBM-roofline kernel the metric names): HIP events on the
                   kernel's own stream over a single-stream leg (2 txn proofs, nothing else on the chip); traffic =
                   algorithmic bytes x the PMC-measured ratio of a tracked profiles/ summary;
  roofline_isolated -- the same kernel alone on the chip at the widest table shape;
  ntt_hbm_gbps  -- BASELINE's second figure: the batched inverse NTT alone at four shapes;
  alu_kernel    -- Merkle leaf hashing (Poseidon; the time-dominant kernels, VALU-issue-bound) over the same leg, as
                   a fraction of the hardware's VALU issue ceiling (SIMDs x clock / 4; instructions per permutation
                   from a tracked SQ-counter summary), next to the same kernels with the chip full, measured live;
  roofline_in_situ  -- (only with --in-situ-profile: the event records cost 2.4 % of the rate) the LDE launches inside
                       the timed region (20 streams share the chip: not the kernel's cost);
  cpu_baseline  -- the oracle (CPU restatement, OpenMP) proving ONE txn of the same block on a warm state.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# One HIP stream per concurrent prover: let the runtime map them onto distinct hardware queues
# (ROCm multiplexes streams onto GPU_MAX_HW_QUEUES = 4 queues by default).  Must be set before
# the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

# SURVEY.md section 8(d) S1 "transfer-txn": range minima of constants.rs:6-18, placeholder widths
S1_LOG_N = (16, 9, 12, 14, 9, 12, 17)
S1_WIDTH = (128, 128, 192, 2432, 512, 320, 16)
SYNTHETIC_REC = dict(rec_air_id=0, rec_n_const=82)   # the recursion-shaped proofs of rounds 1-3 (--synthetic-rec)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md
# HBM traffic / algorithmic bytes of the roofline leg's launches comes from a PMC measurement kept under profiles/
# (two rocprofv3 --pmc passes of this same command, tools/pmc_family_traffic.py).  bench.py cannot collect
# counters itself, so it reads the tracked summary and names it (and the commit it was taken at) beside the
# number; no file, no number.
def tracked(name):
    """the newest tracked counter summary of that name: profiles/r5_<name> when this round's passes were taken, else round 4's"""
    for rnd in ("r5", "r4"):
        p = os.path.join("profiles", "%s_%s" % (rnd, name))
        if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), p)):
            return p
    return os.path.join("profiles", "r4_" + name)


TRAFFIC_FILE = tracked("pmc_lde_family.txt")


# (flag, bp_tune_* entry, help): every run-time knob of the library that bench.py can set
TUNE_KNOBS = (
    ("--merkle-fused", "bp_tune_merkle_fused", "0: one launch per Merkle level"),
    ("--merkle-wide", "bp_tune_merkle_wide", "levels of up to 2^k parents through the wide fused kernel"),
    ("--ntt-split", "bp_tune_ntt_split", "split NTT kernels: 0 auto, 1 never, 2 wherever possible"),
    ("--ntt-mx", "bp_tune_ntt_mx", "matrix-core NTT blocks"),
    ("--poseidon-mx", "bp_tune_poseidon_mx", "0: Poseidon without the matrix cores"),
    ("--poseidon-grouped", "bp_tune_poseidon_grouped", "0 / 2 / 3 groups of partial rounds"),
    ("--k5-spread", "bp_tune_k5_spread", "spread the synthetic AIR's units under load too"),
    ("--quad-threshold-log2", "bp_tune_quad_threshold",
     "hash launches with fewer rows than 2^k take the low-latency Poseidon forms"),
    ("--rec-batch", "bp_tune_rec_batch", "recursion-shaped proofs proved in lock-step per batch (1 = one at a time)"),
    ("--side-lanes", "bp_tune_side_lanes", "trace commitments on idle workers' streams while the device is not loaded (1) or never (0)"),
    ("--host-wait", "bp_tune_host_wait", "0 auto, 1 the runtime's wait, 2 the library's poll-and-sleep wait"),
)


def traffic_ratio():
    import re
    try:
        txt = open(os.path.join(ROOT, TRAFFIC_FILE)).read()
        ratio = float(re.search(r"= ([0-9.]+) x algorithmic", txt).group(1))
        head = re.search(r"HEAD ([0-9a-f]+)", txt)
        return ratio, (head.group(1) if head else None)
    except (OSError, AttributeError, ValueError):
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--txns", type=int, default=256, help="transactions per block")
    ap.add_argument("--threads", type=int, default=0,
                    help="concurrent provers (HIP streams) per GPU; 0 = at most 20, dividing the shard into equally full "
                         "rounds: measured optimum (profiles/r4_sweep_threads.txt); the waits sleep, so the count is not "
                         "tied to host cores")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event kernel timing (roofline leg)")
    ap.add_argument("--in-situ-profile", action="store_true",
                    help="also time every LDE launch INSIDE the timed region with HIP events (roofline_in_situ).  Off by "
                         "default: the event records cost 2.4 %% of the block rate (profiles/r3_order_experiment.txt)")
    ap.add_argument("--no-in-situ-profile", action="store_true", help="(accepted for older command lines: the default)")
    ap.add_argument("--arena-gib", type=float, default=5.0, help="device arena per prover stream")
    # measurement knobs: one table, used for the argument parser, for applying them and for handing EVERY one of them
    # to the --leg-only child, so that the alone-on-the-chip figures of a line come from the configuration its timed
    # region ran with
    for flag, _, help_ in TUNE_KNOBS:
        ap.add_argument(flag, type=int, default=None, help=help_)
    ap.add_argument("--keccak-air", action="store_true",
                    help="every transaction's Keccak table is a real Keccak-f[1600] trace (AIR 1, 2431 columns) instead "
                         "of the 2432-column synthetic table BASELINE's metric is quoted on")
    ap.add_argument("--real-airs", action="store_true",
                    help="the arithmetic, byte-packing, Keccak, Keccak-sponge, logic and memory tables of every transaction are "
                         "proven with the AIRs written from their public definitions (AIR 4, 5, 1, 6, 2, 3: 309 / 299 / 2431 / "
                         "2414 / 524 / 45 columns) instead of synthetic tables of "
                         "BASELINE's widths -- another workload than the metric's, reported as such")
    ap.add_argument("--synthetic-rec", action="store_true",
                    help="the recursion-shaped proofs (22 of a txn's 29 proofs, every aggregation and block proof) are proofs of "
                         "the synthetic AIR on 135 x 82 columns, as in rounds 1-3 (bp_config.rec_air_id = 0), instead of the "
                         "PLONK-shaped circuit that is the default since round 4 (AIR 8: gates by constants, public inputs "
                         "in-circuit, the copy-constraint permutation argument; 85 constant columns, 20 instead of 16 "
                         "auxiliary columns)")
    ap.add_argument("--tree-shape", default="balanced", choices=("balanced", "pairs_then_chain"),
                    help="shape of a shard's aggregation tree (block_driver.aggregation_plan)")
    ap.add_argument("--leg-only", action="store_true",
                    help="(internal) run only the alone-on-the-chip measurements -- single-stream roofline leg, isolated "
                         "LDE, inverse-NTT sweep, Poseidon peak -- and print them as one JSON object; the main run "
                         "starts this as a fresh child process AFTER its timed region")
    ap.add_argument("--phase-marks", action="store_true",
                    help="(measurement knob) write `@phase <name>` lines to stderr for tools/smi_sampler.py")
    ap.add_argument("--leg-skip-extras", action="store_true",
                    help="(with --leg-only, for rocprofv3 passes) only the single-stream leg: no isolated LDE launch, "
                         "NTT sweep or Poseidon peak, so the profiler's per-kernel averages are the leg's own")
    ap.add_argument("--leg-first", action="store_true",
                    help="(measurement knob) round 2's order: the single-stream leg in this process BEFORE the block")
    args = ap.parse_args()
    if args.txns < 2:
        raise SystemExit("--txns must be >= 2 (a block needs an aggregation: decoding.rs:304-347 pads to two)")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python3 bench.py --gpus N` without a launcher: this process has not touched the GPU (no torch.cuda, no
        # bp_* call yet) and starts the N ranks itself as fresh children, relays rank 0's JSON line and exits with
        # the launcher's code.  Under torch.distributed.run (WORLD_SIZE set) this branch is never taken.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # BPG_SHARE_GPU=1 is a rehearsal mode for a 1-GPU box: all ranks use device 0 and the gather runs
    # over gloo (RCCL needs one device per rank).  The driver's real runs never set it.
    if args.threads <= 0:
        # at most 20 streams, and as many as divide the shard into equally full rounds.  Sweep on the 256-txn block
        # (profiles/r4_sweep_threads.txt): 16 / 18 / 20 / 24 streams -> 39.2 / 39.6 / 39.6 / 38.7 txn-proofs/s (flat from 14
        # to 20 since the recursion chains are lock-step batches).  A 32-txn shard runs 16 + 16, a 16-txn shard all 16
        # at once (profiles/r2_shard_streams.txt).
        shard = (args.txns + world - 1) // world
        rounds = (shard + 19) // 20
        args.threads = max(4, (shard + rounds - 1) // rounds)
    share = os.environ.get("BPG_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    if args.leg_only:
        local_rank = int(os.environ.get("BPG_LEG_DEVICE", local_rank))
    # host waits must sleep, not spin (tools/wait_probe.hip): set before torch creates the device context
    import proof_protocol_decoder_amd as _pkg0
    _pkg0.lib().bp_use_blocking_sync(local_rank)
    torch.cuda.set_device(local_rank)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import proof_protocol_decoder_amd as pkg
    from proof_protocol_decoder_amd import proof_gen as pg
    from proof_protocol_decoder_amd.block_driver import BlockDriver, TorchGather, shard_bounds, synthetic_block_irs
    L = pkg.lib()
    L.bp_profile_read.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    for flag, fn, _ in TUNE_KNOBS:
        val = getattr(args, flag[2:].replace("-", "_"))
        if val is not None:
            getattr(L, fn)((1 << val) if flag == "--quad-threshold-log2" else val)

    def read_family(note, leg=True):
        n, ms, by = C.c_uint64(), C.c_double(), C.c_double()
        pkg._lib.check(L.bp_profile_read(0, C.byref(n), C.byref(ms), C.byref(by)))
        if not n.value:
            return None
        ach = by.value / (ms.value * 1e-3) / 1e9
        # the PMC ratio was measured over the single-stream leg's launches: it says nothing about the timed region
        ratio, head = traffic_ratio() if leg else (None, None)
        return {"bound": "hbm", "kernel": "coset-LDE NTT family: ntt16_dit_kernel<12|13|14> + ntt_mx_dit_kernel<13> + ntt_lds_kernel<DIT>", "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                # HBM bytes per launch = algorithmic x the ratio that two rocprofv3 --pmc passes (FETCH_SIZE x2,
                # WRITE_SIZE) of this same command measured over the launches of the single-stream leg
                # (tools/pmc_family_traffic.py; the leg is bracketed by marker dispatches)
                "traffic": round(by.value / n.value * ratio) if ratio else None,
                "traffic_source": ("%s (PMC passes taken at HEAD %s)" % (TRAFFIC_FILE, head)) if ratio else
                                  ("no PMC summary under profiles/ for this code" if leg else
                                   "not measured in the timed region (see roofline.traffic)"),
                "launches": n.value, "avg_launch_us": round(ms.value * 1e3 / n.value, 2),
                "alg_bytes_per_launch": round(by.value / n.value), "note": note}

    def read_leaf_hash():
        # the integer-ALU-bound family (report Gperm/s, claim no HBM fraction)
        n, ms, perms = C.c_uint64(), C.c_double(), C.c_double()
        pkg._lib.check(L.bp_profile_read(2, C.byref(n), C.byref(ms), C.byref(perms)))
        if not n.value:
            return None
        rate = perms.value / (ms.value * 1e-3) / 1e9
        return {"kernel": "Merkle leaf hashing: leaf_hash_mx_kernel<4|2|1> (Poseidon, width 12, MDS layer and the grouped "
                          "partial rounds on the int8 matrix cores)",
                "bound": "int-ALU (VALU issue)", "achieved": round(rate, 3), "unit": "Gperm/s",
                # valu_issue_frac / peak_measured are filled in below from a live measurement of the same kernel
                # family with the chip full (poseidon_peak: 2^21 rows)
                "launches": n.value, "avg_launch_us": round(ms.value * 1e3 / n.value, 2),
                "perms_per_launch": round(perms.value / n.value)}

    HBM_KERNELS = ((3, "fri_fold_kernel", "16 M (1 + 1/16) per layer of M extension values"),
                   (4, "openings_multi_kernel", "8 n C: every coefficient column once"),
                   (5, "fri_combine_*_kernel", "8 n C: every coefficient column once + six result columns"),
                   (6, "aux_suffix_product_kernel", "8 n per column read or written (24 n per product of a synthetic table)"))
    AIR_NAMES = ("synthetic", "keccak_f", "logic", "memory", "arithmetic", "byte_packing", "keccak_sponge", "arithmetic_mul", "plonk")

    def read_hbm_family(fam, kernel, formula):
        n, ms, by = C.c_uint64(), C.c_double(), C.c_double()
        pkg._lib.check(L.bp_profile_read(fam, C.byref(n), C.byref(ms), C.byref(by)))
        if not n.value:
            return None
        ach = by.value / (ms.value * 1e-3) / 1e9
        return {"kernel": kernel, "bound": "hbm", "launches": n.value, "avg_launch_us": round(ms.value * 1e3 / n.value, 2),
                "alg_bytes_per_launch": round(by.value / n.value), "alg_bytes": formula, "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4)}

    def read_k5(only):
        out = {}
        for air in only:
            name = AIR_NAMES[air]
            r = read_hbm_family(7 + air, "quotient_air_kernel<%s>" % name,
                                "8 M (C + A + K + 2): the three LDE matrices read once, two quotient columns written")
            if r:
                if name == "plonk" and r["launches"] % 2 == 0:
                    # two kernels per quotient, both timed into this family: the ten chunk units, then the Poseidon-gate
                    # pass (quotient_plonk_hash_kernel, no algorithmic bytes of its own) -- report per QUOTIENT
                    r.update(kernel="quotient_air_kernel<plonk> + quotient_plonk_hash_kernel (one quotient = both)",
                             launches=r["launches"] // 2, avg_launch_us=round(2 * r["avg_launch_us"], 2),
                             alg_bytes_per_launch=2 * r["alg_bytes_per_launch"])
                r.update(k5_counters().get(name, {}))
                out[name] = r
        return out

    def single_stream_leg():
        """The same workload with ONE prover stream and nothing else on the chip, so every launch of the kernel
        has the device to itself and event time == kernel time (this is what the rocprof summary in profiles/ is
        taken from)."""
        solo = pg.ProverStateBuilder().set(device=local_rank, n_workers=1, arena_bytes=int(args.arena_gib * 2**30),
                                            **(SYNTHETIC_REC if args.synthetic_rec else {})).build()
        solo_driver = BlockDriver(solo, n_threads=1)
        irs = synthetic_block_irs(1000, 2, S1_LOG_N, S1_WIDTH)
        solo_driver.prove_shard(irs[:1])
        # marker dispatches (one-word copies) bracket the leg so that a `rocprofv3 --pmc` pass of this
        # same command can pick out exactly these launches (tools/pmc_family_traffic.py)
        mark = torch.zeros(2, dtype=torch.int64, device="cuda")
        L.bp_debug_copy_u64.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.bp_debug_copy_u64(mark.data_ptr(), mark.data_ptr() + 8, 1, None)
        torch.cuda.synchronize()
        L.bp_profile_reset()
        L.bp_profile_enable(1)
        solo_driver.prove_shard(irs)
        torch.cuda.synchronize()
        L.bp_profile_enable(0)
        L.bp_debug_copy_u64(mark.data_ptr(), mark.data_ptr() + 8, 1, None)
        torch.cuda.synchronize()
        roof = read_family("HIP events on the prover stream (each launch carries its own start / stop events: hipExtLaunchKernelGGL), 2 txn proofs of the same block proved with one stream (no "
                           "co-running kernels); per txn 7 table proofs, 3 lock-step batches of 7 recursion-shaped proofs, a root proof")
        alu = read_leaf_hash()
        # the other HBM-class kernels SURVEY.md section 8(d) names, over the same leg, and K5 on the synthetic tables
        others = {name: r for fam, name, formula in HBM_KERNELS for r in [read_hbm_family(fam, name, formula)] if r}
        k5 = read_k5((0, 8))
        if not args.leg_skip_extras:
            # the same leg with the six tables that have an AIR proven with it: K5 per air_id
            irs_r = synthetic_block_irs(1001, 2, S1_LOG_N, S1_WIDTH, **REAL_AIRS)
            solo_driver.prove_shard(irs_r[:1])
            L.bp_profile_reset()
            L.bp_profile_enable(1)
            solo_driver.prove_shard(irs_r)
            torch.cuda.synchronize()
            L.bp_profile_enable(0)
            k5.update(read_k5((1, 2, 3, 4, 5, 6)))
        solo_driver.close()
        solo.close()
        return roof, alu, others, k5

    def side_block(real_airs, synthetic_rec):
        """The driver's one command never touches AIR 1..6 outside pytest: a 64-txn block of another workload than the
        metric's, 16 prover streams, here in the child process.  real_airs: the six tables with an AIR are proven with
        it (the sponge table's rows look their permutations up in the Keccak-f table); synthetic_rec: the
        recursion-shaped proofs are proofs of the synthetic AIR (bp_config.rec_air_id = 0, the workload of rounds 1-3)."""
        n, thr = 64, 16
        rec = SYNTHETIC_REC if synthetic_rec else {}
        st = pg.ProverStateBuilder().set(device=local_rank, n_workers=thr, arena_bytes=int(args.arena_gib * 2**30), **rec).build()
        drv = BlockDriver(st, n_threads=thr)
        blocks = [synthetic_block_irs(3000 + b, n, S1_LOG_N, S1_WIDTH, **(REAL_AIRS if real_airs else {})) for b in range(3)]
        last = drv.prove_block_distributed(blocks[0], 0, 1, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in blocks[1:]:
            last = drv.prove_block_distributed(b, 0, 1, None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pg.VerifierState.from_prover_state(st).verify(last)
        drv.close()
        st.close()
        what = []
        if real_airs:
            what.append("the six tables that have an AIR proven with it (AIR 4, 5, 1, 6, 2, 3: 309 / 299 / 2431 / 2414 / 524 / 45 "
                        "columns), the CPU table synthetic; cross-table lookups keccak_sponge -> keccak_f, keccak_sponge -> logic and byte_packing -> memory checked in every txn")
        if synthetic_rec:
            what.append("every recursion-shaped proof a proof of the synthetic AIR (135 x 82 columns, 16 auxiliary columns) "
                        "instead of the PLONK-shaped circuit: the recursion workload of rounds 1-3")
        else:
            what.append("recursion-shaped proofs on the PLONK-shaped circuit (AIR 8), as in the metric's run")
        return {"value": round(n * 2 / dt, 3), "unit": "txn-proofs/s", "steps": 2, "warmup": 1, "prover_streams": thr,
                "workload": "64-txn block, " + "; ".join(what) + "; another workload than the metric's"}

    if args.leg_only:
        # child mode: everything that is measured alone on the chip, in a process of its own
        roof, alu, others, k5 = single_stream_leg()
        out = {"roofline": roof, "alu_kernel": alu, "hbm_kernels": others, "k5": k5}
        if not args.leg_skip_extras:
            out.update(real_airs=side_block(True, False), synthetic_rec=side_block(False, True))
            out.update(roofline_isolated=isolated_roofline(pkg, torch), ntt_hbm_gbps=ntt_gbps(pkg, torch))
            if alu:
                finish_alu_kernel(alu, poseidon_peak(pkg, torch))
        print(json.dumps(out), flush=True)
        return

    if args.phase_marks and rank == 0:
        pr = torch.cuda.get_device_properties(local_rank)
        print("@pci %04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id), file=sys.stderr, flush=True)

    def phase(name):
        if args.phase_marks and rank == 0:
            free_b, total_b = torch.cuda.mem_get_info()
            print("@phase %s (device memory free %.1f of %.1f GiB)" % (name, free_b / 2**30, total_b / 2**30),
                  file=sys.stderr, flush=True)

    roofline = alu_kernel = None
    if args.leg_first and not args.no_profile and rank == 0:
        phase("single-stream leg")
        roofline, alu_kernel, _, _ = single_stream_leg()
    phase("state build")

    t_build = time.time()
    # ProverStateBuilder::default() ranges (constants.rs:6-18), as the reference builds them
    rec = SYNTHETIC_REC if args.synthetic_rec else {}
    state = pg.ProverStateBuilder().set(device=local_rank, n_workers=args.threads,
                                         arena_bytes=int(args.arena_gib * 2**30), **rec).build()
    t_build = time.time() - t_build
    driver = BlockDriver(state, n_threads=args.threads, tree_shape=args.tree_shape)
    gather = TorchGather(torch.device("cpu") if share else torch.device("cuda", local_rank)) if world > 1 else None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    blocks = [synthetic_block_irs(b, args.txns, S1_LOG_N, S1_WIDTH, keccak_air=args.keccak_air or args.real_airs,
                                  logic_air=args.real_airs, memory_air=args.real_airs, arithmetic_air=args.real_airs,
                                  byte_packing_air=args.real_airs, keccak_sponge_air=args.real_airs)
              for b in range(args.warmup + args.steps)]
    last = None
    phase("warmup")
    for b in range(args.warmup):
        last = driver.prove_block_distributed(blocks[b], rank, world, gather)
    in_situ = args.in_situ_profile and not args.no_profile and not args.no_in_situ_profile
    if in_situ:
        L.bp_profile_reset()
        L.bp_profile_enable(1)
    barrier()
    step_ms = []
    clock = ClockSampler(torch, local_rank) if rank == 0 else None   # engine clock actually held in the timed region
    t0 = time.perf_counter()
    for b in range(args.warmup, args.warmup + args.steps):
        phase("timed step %d" % (b - args.warmup))
        t_s = time.perf_counter()
        last = driver.prove_block_distributed(blocks[b], rank, world, gather)   # returns with the block proof
        step_ms.append(round((time.perf_counter() - t_s) * 1e3, 1))
    barrier()
    dt = time.perf_counter() - t0
    clock_held = clock.stop() if clock else None
    phase("after the timed region")
    L.bp_profile_enable(0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    roofline_in_situ = None
    if in_situ:
        roofline_in_situ = read_family("HIP events around every launch in the timed region; %d prover streams share "
                                       "the chip, mostly with integer-ALU-bound Poseidon kernels, so a launch's "
                                       "duration is not the kernel's own cost" % args.threads, leg=False)

    if rank == 0:
        # acceptance: the block proof verifies (VerifierState::verify, verifier_state.rs:56-71)
        pg.VerifierState.from_prover_state(state).verify(last)
    L.bp_host_wait_mode.restype = C.c_int
    host_waits = {1: "sleep (hipDeviceScheduleBlockingSync)",
                  2: "device already in use: mode left alone, the library's poll-and-sleep wait"}.get(
        L.bp_host_wait_mode(local_rank), "undecided")
    t_build_info = {"state_build_s": round(t_build, 2), "state_device_gib": round(state.device_bytes / 2**30, 2),
                    "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "state_warnings": state.warnings or None}
    driver.close()
    state.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    out = {
        "metric": "txn-proofs/sec (whole node), %d-txn synthetic block" % args.txns,
        "value": round(args.txns * args.steps / dt, 3), "unit": "txn-proofs/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3 / args.steps, 2),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64 (Goldilocks field)",
        "data": "synthetic",
        "config": {"workload": "%d-txn synthetic block, S1 transfer-txn tables logN=%s widths=%s, 7 table STARKs "
                               "+ 22 recursion-shaped proofs per txn, %d agg proofs + 1 block proof per block"
                               % (args.txns, list(S1_LOG_N), list(S1_WIDTH), args.txns - 1),
                   "keccak_table": "Keccak-f[1600] AIR, 2431 columns" if (args.keccak_air or args.real_airs)
                                   else "synthetic AIR, 2432 columns",
                   "recursion_proofs": "synthetic AIR, 135 columns, 82 constants" if args.synthetic_rec
                                       else "PLONK-shaped circuit (AIR 8), 135 wires, 85 constants, 20 auxiliary columns; the public-input list is hashed in-circuit by Poseidon-gate rows, and every recursion circuit walks one Merkle path per child proof in-circuit (the aggregation / block circuits from the opened row itself)",
                   **({"logic_table": "logic AIR, 524 columns", "memory_table": "memory AIR, 45 columns",
                       "arithmetic_table": "arithmetic AIR, 309 columns",
                       "byte_packing_table": "byte-packing AIR, 299 columns",
                       "keccak_sponge_table": "Keccak sponge AIR, 2414 columns"} if args.real_airs else {}),
                   "txns_per_block": args.txns, "prover_streams_per_gpu": args.threads,
                   "sharding": "contiguous txn slices; the %d sub-block proofs meet in a pairwise tree over ranks (RCCL send / recv: "
                               "rank 2k+1 -> 2k, 4k+2 -> 4k, ...)" % world,
                   "scheduler": "bp_prove_shard (csrc/gi.cpp): the slice's txn proofs and its aggregation tree on the library's own "
                                "thread pool, aggregations ahead of waiting transactions",
                   "ms_of_each_step_rank0": step_ms, "host_waits": host_waits, **t_build_info},
    }
    out["engine_clock"] = clock_held
    out["valu_issue"] = block_valu_issue(out["value"], clock_held, synthetic_rec=args.synthetic_rec, real_airs=args.real_airs or args.keccak_air)
    alone = {}
    if not args.no_profile:
        if args.leg_first:
            alone = {"roofline": roofline, "alu_kernel": alu_kernel}
            if world == 1:
                alone["roofline_isolated"] = isolated_roofline(pkg, torch)
                alone["ntt_hbm_gbps"] = ntt_gbps(pkg, torch)
                if alu_kernel:
                    finish_alu_kernel(alu_kernel, poseidon_peak(pkg, torch))
        else:
            # Everything that is measured ALONE on the chip runs after the timed region, in a fresh child process:
            # this process's prover state is closed (its 126 GiB are back), every other rank has passed the
            # barrier above, and the timed region itself always starts on a device nothing has run on -- so
            # `value` is the same number with and without --no-profile.
            import subprocess
            env = {k: v for k, v in os.environ.items()
                   if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
            env["BPG_LEG_DEVICE"] = str(local_rank)
            knobs = ["--arena-gib", str(args.arena_gib)] + (["--synthetic-rec"] if args.synthetic_rec else [])
            for flag, _, _ in TUNE_KNOBS:
                val = getattr(args, flag[2:].replace("-", "_"))
                if val is not None:
                    knobs += [flag, str(val)]
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg-only"] + knobs, env=env,
                               stdout=subprocess.PIPE, text=True)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                raise SystemExit("the --leg-only child failed (exit %d)" % r.returncode)
            alone = json.loads(lines[-1])
    out["roofline"] = alone.get("roofline")
    out["roofline_in_situ"] = roofline_in_situ
    out["alu_kernel"] = alone.get("alu_kernel")
    for k in ("hbm_kernels", "k5", "real_airs", "synthetic_rec", "roofline_isolated", "ntt_hbm_gbps"):
        if k in alone:
            out[k] = alone[k]
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(blocks[args.warmup][0], synthetic_rec=args.synthetic_rec)
    print(json.dumps(out), flush=True)


# Hardware ceiling of a VALU-issue-bound kernel: 256 CUs x 4 SIMDs, one wave64 VALU instruction per SIMD every
# 4 cycles (a SIMD has 16 lanes), at the 2.4 GHz peak engine clock (MI355X_MICROARCH.md).
N_SIMD, PEAK_CLOCK_HZ = 1024, 2.4e9
VALU_PEAK_WAVE_INSTS_PER_S = N_SIMD * PEAK_CLOCK_HZ / 4
class ClockSampler:
    """The engine clock (sclk) of this rank's card, read from sysfs (hwmon freq1_input of the card with the device's PCI
    address) every 50 ms by a thread while the timed region runs: the clock the chip actually HELD under this load --
    the VALU-issue ceiling is SIMDs x that clock / cycles per instruction, not x the 2.4 GHz of the data sheet."""

    def __init__(self, torch, device):
        import threading
        self.samples, self._stop, self.path = [], threading.Event(), None
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from smi_sampler import find_sources
            p = torch.cuda.get_device_properties(device)
            bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
            self.path = find_sources(bdf).get("sclk_Hz")
        except Exception:
            self.path = None
        if not self.path:
            self.thread = None
            return

        def run():
            while not self._stop.is_set():
                try:
                    self.samples.append(int(open(self.path).read().strip()))
                except (OSError, ValueError):
                    pass
                self._stop.wait(0.05)
        self.thread = threading.Thread(target=run, daemon=True)
        self.thread.start()

    def stop(self):
        if not self.thread:
            return None
        self._stop.set()
        self.thread.join()
        hz = [x for x in self.samples if x > 0]
        if not hz:
            return None
        return {"sclk_mhz_mean": round(sum(hz) / len(hz) / 1e6, 1), "sclk_mhz_min": round(min(hz) / 1e6, 1),
                "sclk_mhz_max": round(max(hz) / 1e6, 1), "samples": len(hz), "source": "hwmon freq1_input, every 50 ms over the timed region"}


LOADED_SQ_FILE = tracked("sq_loaded_by_kernel.txt")
CLASS_MIX_FILE = os.path.join("profiles", "r5_valu_class_mix.txt")


def block_valu_issue(rate, clock_held, synthetic_rec=False, real_airs=False):
    """How close the block rate is to the chip's VALU-issue ceiling: (wave-instructions per txn proof, SQ counters of the
    loaded run) x rate against 1024 SIMDs x clock / cycles per instruction -- once the data-sheet way (2.4 GHz, a flat
    4 cycles), once with the clock held in this run's timed region and the cycle cost weighted by the instruction
    classes of the kernels that issue them (tools/valu_class_mix.py: the cheap 32-bit class at 2.45 cycles, the carry /
    multiply / 64-bit class at 4.5: profiles/r2_issue_rates.txt)."""
    import re
    if synthetic_rec or real_airs:
        return None   # the tracked counters are those of the default workload
    try:
        txt = open(os.path.join(ROOT, LOADED_SQ_FILE)).read()
        n_txn = int(re.search(r"--txns (\d+)", txt).group(1))
        per_txn = sum(float(m.group(1)) for m in re.finditer(r"VALU ([0-9.]+e[+0-9]+) \(", txt)) / n_txn
    except (OSError, AttributeError, ValueError):
        return None
    if not per_txn:
        return None
    out = {"valu_wave_insts_per_txn": per_txn, "counters_source": LOADED_SQ_FILE,
           "frac_of_peak_clock_flat_4_cycles": round(per_txn * rate / VALU_PEAK_WAVE_INSTS_PER_S, 3)}
    try:
        mix = open(os.path.join(ROOT, CLASS_MIX_FILE)).read()
        cyc = float(re.search(r"weighted cycles per VALU instruction \(block mix\): ([0-9.]+)", mix).group(1))
        out["weighted_cycles_per_inst"] = cyc
        out["class_mix_source"] = CLASS_MIX_FILE
        if clock_held:
            hz = clock_held["sclk_mhz_mean"] * 1e6
            out["frac_at_held_clock_weighted"] = round(per_txn * rate / (N_SIMD * hz / cyc), 3)
            out["frac_at_held_clock_flat_4_cycles"] = round(per_txn * rate / (N_SIMD * hz / 4), 3)
    except (OSError, AttributeError, ValueError):
        pass
    return out


SQ_FILE = tracked("hash_sq_counters.txt")


def valu_insts_per_perm():
    """VALU wave-instructions per Poseidon permutation of the shipped leaf-hash kernel (four sets per wave, grouped
    partial rounds), from the tracked SQ-counter summary (rocprofv3 --pmc SQ_INSTS_VALU over 2^21 rows x 8
    permutations; tools/prof_round4.sh)."""
    import re
    try:
        txt = open(os.path.join(ROOT, SQ_FILE)).read()
        blk = txt[txt.index("leaf_hash_mx_kernel<4, three groups>"):]
        insts = float(re.search(r"SQ_INSTS_VALU\s+([0-9.e+]+)", blk).group(1))
        mfma = float(re.search(r"SQ_INSTS_MFMA\s+([0-9.e+]+)", blk).group(1))
        m = re.search(r"= ([0-9]+) permutations", txt)
        perms = float(m.group(1)) if m else float((1 << 21) * 8)
        return insts / perms, mfma / perms
    except (OSError, AttributeError, ValueError):
        return None, None


K5_FILE = tracked("k5_counters.txt")
REAL_AIRS = dict(keccak_air=True, logic_air=True, memory_air=True, arithmetic_air=True, byte_packing_air=True,
                 keccak_sponge_air=True)


def k5_counters():
    """Per AIR, from the tracked counter summary of tools/k5_air_probe.py under rocprofv3 (tools/prof_round4.sh):
    VALU lane-instructions per constraint evaluation and HBM bytes fetched per algorithmic byte read."""
    import re
    out = {}
    try:
        for line in open(os.path.join(ROOT, K5_FILE)):
            m = re.match(r"(\w+)\s+.*valu_per_constraint=([0-9.]+).*fetch_over_algorithmic=([0-9.]+)", line)
            if m:
                out[m.group(1)] = {"valu_lane_insts_per_constraint": float(m.group(2)),
                                   "hbm_fetch_over_algorithmic_read": float(m.group(3)), "counters_source": K5_FILE}
    except OSError:
        pass
    return out


def finish_alu_kernel(alu, peak):
    """Anchor the Poseidon family to the hardware's VALU issue rate (not to its own best run): wave-instructions
    per second = Gperm/s x instructions per permutation (SQ counters, tracked file) against SIMDs x clock / 4."""
    per_perm, mfma_per_perm = valu_insts_per_perm()
    alu["peak_measured"] = peak
    alu["frac_of_peak_measured"] = round(alu["achieved"] / peak, 3)
    if per_perm:
        alu["valu_insts_per_perm"] = round(per_perm, 1)
        alu["mfma_insts_per_perm"] = round(mfma_per_perm, 2)
        alu["valu_insts_source"] = SQ_FILE
        alu["valu_issue_peak_insts_per_s"] = VALU_PEAK_WAVE_INSTS_PER_S
        alu["valu_issue_frac"] = round(alu["achieved"] * 1e9 * per_perm / VALU_PEAK_WAVE_INSTS_PER_S, 3)
        alu["valu_issue_frac_chip_full"] = round(peak * 1e9 * per_perm / VALU_PEAK_WAVE_INSTS_PER_S, 3)
    # what the kernel asks of the matrix cores at the chip-full rate: MFMA wave-instructions per permutation (SQ
    # summary; 11.25 = 30 rounds x 6 per 16 states before the grouped rounds) x 32768 int8 ops each.  Informational: the
    # kernel is bound by VALU issue, and most of these multiplies are by the zeros of plane-diagonal or sparse operands.
    INT8_DENSE_PEAK_TOPS = 5000.0   # MI355X_MICROARCH.md: I8 = 2 x the BF16 rate per clock
    tops = peak * 1e9 * (mfma_per_perm or 11.25) * 32768 / 1e12
    alu["mfma_int8"] = {"achieved_at_peak_rate": round(tops, 1), "peak": INT8_DENSE_PEAK_TOPS,
                        "unit": "TOP/s", "frac": round(tops / INT8_DENSE_PEAK_TOPS, 3)}


def isolated_roofline(pkg, torch):
    """The LDE kernel alone on the chip at the widest table shape (2^14 x 2432, rate 2)."""
    log_n, C_, r = 14, 2432, 1
    n = 1 << log_n
    coeffs = torch.randint(0, 2**62, (C_, n), dtype=torch.int64, device="cuda")
    pkg.ops.lde_batch(coeffs, r, from_coeffs=True)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        pkg.ops.lde_batch(coeffs, r, from_coeffs=True)   # launches on torch's current stream
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    alg = 8.0 * n * C_ * (1 + (1 << r))
    ach = alg / (best * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "ntt16_dit_kernel<14> (coset LDE), 2^14 x 2432, rate 2, alone", "achieved": round(ach, 1),
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "launch_ms": round(best, 4)}


def poseidon_peak(pkg, torch):
    """Gperm/s of the Merkle commitment (leaf hashing + levels) of a 2^20 x 64 table at rate 2, alone on the chip:
    2^21 rows x 8 permutations, 32768 waves of the four-set matrix-core kernel -- every SIMD issuing all the time."""
    log_n, cols, r = 20, 64, 1
    lde = torch.randint(0, 2**62, (cols, 1 << (log_n + r)), dtype=torch.int64, device="cuda")
    pkg.ops.merkle_commit(lde, log_n, r, 4)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        pkg.ops.merkle_commit(lde, log_n, r, 4)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    perms = (1 << (log_n + r)) * ((cols + 7) // 8) + (1 << (log_n + r))
    return round(perms / (best * 1e-3) / 1e9, 3)


def ntt_gbps(pkg, torch):
    """BASELINE metric (ii), Goldilocks NTT HBM GB/s: the batched inverse NTT (natural -> bit-reversed, out of
    place: bp_intt_batch; one kernel launch up to 2^14 rows, one more HBM round trip up to 2^22) alone on the chip
    at four shapes of SURVEY.md section 8(d) S4.
    Algorithmic bytes 16*n*C per transform; fraction of the 8 TB/s HBM peak beside it."""
    out = {}
    for log_n, cols in ((12, 2048), (14, 2048), (16, 256), (20, 64)):
        v = torch.randint(0, 2**62, (cols, 1 << log_n), dtype=torch.int64, device="cuda")
        o = torch.empty_like(v)
        pkg.ops.intt_batch(v, o)      # values -> coefficients, out of place as in PolynomialBatch::from_values
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            pkg.ops.intt_batch(v, o)
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        gbps = 16.0 * (1 << log_n) * cols / (best * 1e-3) / 1e9
        out["2^%d x %d" % (log_n, cols)] = {"GB/s": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4)}
        del v
    return out


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return min(n, 64)


def cpu_baseline(ir, synthetic_rec=False):
    """Oracle (CPU restatement, OpenMP over the host cores) proving one txn of the block."""
    from oracle import pyoracle  # checker / baseline only
    pyoracle.build()
    cores = usable_cores()
    try:  # OpenMP would otherwise start one thread per host core, not per core we may run on
        C.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except OSError:
        pass
    lo, hi = list(S1_LOG_N), [x + 1 for x in S1_LOG_N]
    st = pyoracle.PgState(table_log_lo=lo, table_log_hi=hi, stark_rate_bits=1, stark_cap_height=4,
                          stark_num_queries=84, stark_pow_bits=16, arity_bits=4, final_poly_bits=5, rec_log_n=13,
                          rec_n_cols=135, rec_n_const=82 if synthetic_rec else 85, rec_rate_bits=3, rec_num_queries=28, rec_pow_bits=16,
                          shrink_depth=3, rec_air_id=0 if synthetic_rec else 8)
    import struct
    words = list(struct.unpack("<25Q", ir.to_bytes()))
    # like for like with the GPU figure, which excludes bp_state_build: the eight circuits this txn touches are
    # preprocessed BEFORE the clock starts (the oracle builds circuits lazily) and that time is reported beside it
    t0 = time.perf_counter()
    st.preprocess(words)
    t_pre = time.perf_counter() - t0
    t0 = time.perf_counter()
    st.txn(words)
    dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 4), "unit": "txn-proofs/s", "cores": cores, "kind": "port",
            "sample": "1 txn proof of the same block (7 tables + 22 recursion-shaped proofs) on a warm state: %.1f s; "
                      "preprocessing of the 8 circuits it touches, not in the figure: %.1f s" % (dt, t_pre),
            "seconds_per_txn": round(dt, 2), "preprocess_s_excluded": round(t_pre, 2),
            "value_incl_preprocessing": round(1.0 / (dt + t_pre), 4)}


if __name__ == "__main__":
    main()
