"""K5 alone per AIR: bp_quotient_eval(air_id, ...) on random LDE matrices at the tables' S1 heights, alone on the chip
and (bp_tune_assume_loaded(1)) in the one-pass form the library uses under load.  Reports time per launch, rows/s and
constraint evaluations per second -- what a real AIR costs next to the synthetic one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proof_protocol_decoder_amd as bpg

L = bpg.lib()
P = 0xFFFFFFFF00000001
CASES = [  # name, air_id, columns, log_n (the S1 height of the table), n_const
    ("synthetic 128 cols (S1 arithmetic width)", 0, 128, 16, 0),
    ("synthetic 2432 cols (S1 keccak width)", 0, 2432, 14, 0),
    ("arithmetic (AIR 4)", 4, 309, 16, 0),
    ("byte packing (AIR 5)", 5, 297, 9, 0),
    ("byte packing (AIR 5) at 2^14", 5, 297, 14, 0),
    ("keccak_f (AIR 1)", 1, 2430, 14, 0),
    ("logic (AIR 2)", 2, 523, 12, 0),
    ("logic (AIR 2) at 2^16", 2, 523, 16, 0),
    ("memory (AIR 3)", 3, 44, 17, 0),
    ("keccak sponge (AIR 6)", 6, 2414, 9, 0),
    ("keccak sponge (AIR 6) at 2^12", 6, 2414, 12, 0),
    ("multiplication (AIR 7)", 7, 1217, 14, 0),
]
g = torch.Generator(device="cuda").manual_seed(1)
for name, air, C_, log_n, K in CASES:
    rows = (1 << log_n) << 1
    tr = torch.randint(0, 2**62, (C_, rows), dtype=torch.int64, device="cuda", generator=g)
    aux = torch.randint(0, 2**62, (C_ // 8, rows), dtype=torch.int64, device="cuda", generator=g)
    d = bpg.ops.air_describe(air, n_cols=C_) if air == 0 else bpg.ops.air_describe(air)
    n_cons = d.n_air_constraints + d.n_ctl_constraints
    cfg = bpg.ops.stark_cfg(log_n, C_)
    out = []
    for loaded in (0, 1):
        L.bp_tune_assume_loaded(loaded)
        bpg.ops.quotient_eval(cfg, tr, aux, None, (3, 5, 7, 11), (13, 17), air_id=air)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            bpg.ops.quotient_eval(cfg, tr, aux, None, (3, 5, 7, 11), (13, 17), air_id=air)
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        out.append(best)
    L.bp_tune_assume_loaded(-1)
    print("%-42s 2^%d x %d, %4d constraints: %8.1f us spread over workgroup rows, %8.1f us in the form taken under load (one pass for the synthetic AIR, 256 workgroups for the others); %6.2f G constraint "
          "evaluations/s, %5.1f ns per row" % (name, log_n, C_, n_cons, out[0] * 1e3, out[1] * 1e3,
                                                n_cons * rows / (min(out) * 1e-3) / 1e9, min(out) * 1e6 / rows), flush=True)
    del tr, aux
