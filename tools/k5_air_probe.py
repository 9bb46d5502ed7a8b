"""K5 alone per AIR: bp_quotient_eval(air_id, ...) on random LDE matrices at the tables' S1 heights, alone on the chip
and (bp_tune_assume_loaded(1)) in the form the library uses under load.  Reports time per launch, algorithmic GB/s
(8 M (C + A + 2): the LDE matrices read once, two quotient columns written), constraint
evaluations per second -- what a real AIR costs next to the synthetic one.
  python tools/k5_air_probe.py             the timing table
  python tools/k5_air_probe.py --counters  ONE launch per AIR (spread form, S1 height) for a rocprofv3 --pmc pass
                                           (tools/prof_round4.sh; summary by tools/prof_round4_summaries.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proof_protocol_decoder_amd as bpg

L = bpg.lib()
COUNTERS = "--counters" in sys.argv
CASES = [  # name, air_id, columns, log_n (the S1 height of the table), in the counters pass
    ("synthetic 128 cols (S1 arithmetic width)", 0, 128, 16, False),
    ("synthetic 2432 cols (S1 keccak width)", 0, 2432, 14, True),
    ("arithmetic (AIR 4)", 4, 309, 16, True),
    ("byte packing (AIR 5)", 5, 299, 9, False),
    ("byte packing (AIR 5) at 2^14", 5, 299, 14, True),
    ("keccak_f (AIR 1)", 1, 2431, 14, True),
    ("logic (AIR 2)", 2, 524, 12, False),
    ("logic (AIR 2) at 2^16", 2, 524, 16, True),
    ("memory (AIR 3)", 3, 45, 17, True),
    ("keccak sponge (AIR 6)", 6, 2414, 9, False),
    ("keccak sponge (AIR 6) at 2^12", 6, 2414, 12, True),
    ("multiplication (AIR 7)", 7, 1217, 14, True),
    ("plonk (AIR 8), one recursion-shaped proof", 8, 135, 13, True),
    ("synthetic 135 x 82 at the recursion shape", 0, 135, 13, False),
]
REC = dict(n_const=82, deg_pow=3, rate_bits=3)   # the recursion shape: rate 8; AIR 8 has 85 constant columns
g = torch.Generator(device="cuda").manual_seed(1)
for name, air, C_, log_n, in_counters in CASES:
    if COUNTERS and not in_counters:
        continue
    rec = dict(REC, n_const=85 if air == 8 else 82) if log_n == 13 and C_ == 135 else None
    rows = (1 << log_n) << (3 if rec else 1)
    d = (bpg.ops.air_describe(air, n_cols=C_, **({k: rec[k] for k in ("n_const", "deg_pow")} if rec else {})) if air == 0
         else bpg.ops.air_describe(air))
    tr = torch.randint(0, 2**62, (C_, rows), dtype=torch.int64, device="cuda", generator=g)
    aux = torch.randint(0, 2**62, (d.n_aux, rows), dtype=torch.int64, device="cuda", generator=g)
    cst = torch.randint(0, 2**62, (rec["n_const"], rows), dtype=torch.int64, device="cuda", generator=g) if rec else None
    n_cons = d.n_air_constraints + d.n_ctl_constraints
    alg = 8.0 * rows * (C_ + d.n_aux + (rec["n_const"] if rec else 0) + 2)
    cfg = bpg.ops.stark_cfg(log_n, C_, **rec) if rec else bpg.ops.stark_cfg(log_n, C_)
    if COUNTERS:
        L.bp_tune_assume_loaded(0)
        bpg.ops.quotient_eval(cfg, tr, aux, cst, (3, 5, 7, 11), (13, 17), air_id=air)
        torch.cuda.synchronize()
        print("counters case: air %d %s rows %d cols %d aux %d constraints %d alg_bytes %d" % (air, d.name.decode(), rows, C_, d.n_aux, n_cons, alg), flush=True)
        continue
    out = []
    for loaded in (0, 1):
        L.bp_tune_assume_loaded(loaded)
        bpg.ops.quotient_eval(cfg, tr, aux, cst, (3, 5, 7, 11), (13, 17), air_id=air)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            bpg.ops.quotient_eval(cfg, tr, aux, cst, (3, 5, 7, 11), (13, 17), air_id=air)
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        out.append(best)
    L.bp_tune_assume_loaded(-1)
    print("%-44s 2^%d x %d (+%d aux), %4d constraints: %8.1f us spread over workgroup rows (%6.1f GB/s algorithmic), %8.1f us in the form taken "
          "under load (one pass for the synthetic AIR, 256 workgroups for the others); %6.2f G constraint evaluations/s, %5.1f ns per row"
          % (name, log_n, C_, d.n_aux, n_cons, out[0] * 1e3, alg / (out[0] * 1e-3) / 1e9, out[1] * 1e3,
             n_cons * rows / (min(out) * 1e-3) / 1e9, min(out) * 1e6 / rows), flush=True)
    del tr, aux, cst
