"""NTT sweep, log N in 12..24, C in {64, 256, 2048} where it fits in HBM.
Prints algorithmic GB/s (16*N*C per iNTT; 8*N*C*(2+2^r) for values -> coefficients + LDE, r = 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proof_protocol_decoder_amd as bpg

def timeit(f, reps=4):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best

print("| log N | C | iNTT ms | iNTT GB/s | frac of 8 TB/s | iNTT+LDE(r=1) ms | GB/s (alg) | frac |")
print("|---|---|---|---|---|---|---|---|")
for log_n in (12, 14, 16, 18, 20, 22, 24):
    for C in (64, 256, 2048):
        n = 1 << log_n
        if n * C * 8 * 4.5 > 200e9:   # values + coeffs + 2x LDE must fit comfortably
            continue
        v = torch.randint(0, 2**62, (C, n), dtype=torch.int64, device="cuda")
        t1 = timeit(lambda: bpg.ops.ntt_batch_(v, bpg.ops.NTT_INV_NAT2BR))
        t2 = timeit(lambda: bpg.ops.lde_batch(v, 1), reps=2)
        g1 = 16 * n * C / t1 / 1e6
        g2 = 8 * n * C * 4 / t2 / 1e6
        print("| %d | %d | %.3f | %.0f | %.3f | %.3f | %.0f | %.3f |" % (log_n, C, t1, g1, g1 / 8000, t2, g2, g2 / 8000), flush=True)
        del v
        torch.cuda.empty_cache()
