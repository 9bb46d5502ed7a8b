"""Per-shape times of the coset-LDE launches that make up bench.py's single-stream roofline leg (alone on the chip):
the recursion shape's four launches (2^13 points, rate 8, 135 / 16 / 16 / 6 columns) and the S1 tables, for the kernel
forms the planner can take (VALU butterflies, matrix-core passes, split).  python tools/ntt_leg_shapes_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proof_protocol_decoder_amd as bpg

L = bpg.lib()


def timeit(f, reps=9):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best * 1e3   # us


shapes = [(13, 135, 3), (13, 16, 3), (13, 6, 3), (16, 128, 1), (16, 16, 1), (14, 2432, 1), (14, 304, 1), (17, 16, 1), (12, 192, 1), (9, 128, 1)]
forms = [("valu", 0, 0), ("valu+split2", 0, 2), ("valu+split3", 0, 3), ("mx(default 3)", 3, 0), ("mx2", 2, 0), ("mx2+split3", 2, 3)]
print("%-16s" % "shape" + "".join("%16s" % f[0] for f in forms) + "   (us per launch | GB/s of algorithmic bytes)")
for log_n, C, r in shapes:
    n = 1 << log_n
    v = torch.randint(0, 2**62, (C, n), dtype=torch.int64, device="cuda")
    alg = 8 * n * C * (1 + (1 << r))
    row = "2^%-2d x %-4d r%d   " % (log_n, C, 1 << r)
    for name, mx, split in forms:
        L.bp_tune_ntt_mx(mx)
        L.bp_tune_ntt_split(split)
        t = timeit(lambda: bpg.ops.lde_batch(v, r, from_coeffs=True))
        row += "%8.1f|%6.0f " % (t, alg / t / 1e3)
    print(row, flush=True)
L.bp_tune_ntt_mx(3)
L.bp_tune_ntt_split(0)
