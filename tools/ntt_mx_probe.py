"""A/B of the NTT block kernels: VALU butterflies against the matrix-core passes, per shape and per resident-workgroup
setting (python tools/ntt_mx_probe.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proof_protocol_decoder_amd as bpg


def timeit(f, reps=7):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


L = bpg.lib()
for name, log_n, C, r in [("sweep12", 12, 2048, 1), ("keccak", 14, 2432, 1), ("recursion", 13, 135, 3), ("rec-aux", 13, 16, 3),
                          ("arith", 16, 128, 1), ("sweep20", 20, 64, 1)]:
    n = 1 << log_n
    v = torch.randint(0, 2**62, (C, n), dtype=torch.int64, device="cuda")
    o = torch.empty_like(v)
    row = "%-10s 2^%d x %d r=%d |" % (name, log_n, C, r)
    for label, mx, per in [("valu", 0, 0), ("mx", 2, 0), ("mx/1", 2, 1), ("mx/2", 2, 2), ("mx/3", 2, 3), ("mx/4", 2, 4)]:
        L.bp_tune_ntt_mx(mx)
        L.bp_tune_ntt_mx_wg_per_cu(per)
        t_i = timeit(lambda: bpg.ops.intt_batch(v, o))
        t_l = timeit(lambda: bpg.ops.lde_batch(v, r, from_coeffs=True))
        row += " %s intt %5.0f lde %5.0f |" % (label, 16 * n * C / t_i / 1e6, 8 * n * C * (1 + (1 << r)) / t_l / 1e6)
    print(row, flush=True)
L.bp_tune_ntt_mx(3)
L.bp_tune_ntt_mx_wg_per_cu(0)
