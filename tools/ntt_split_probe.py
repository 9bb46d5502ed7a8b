"""A/B of the split (two workgroups per block) NTT kernels: bp_tune_ntt_split 1 = never, 2 = wherever possible,
0 = automatic, on the launch shapes of the synthetic txn proof.  Prints ms and algorithmic GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proof_protocol_decoder_amd as bpg

L = bpg.lib()


def timeit(f, reps=7):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


shapes = [("rec trace", 13, 135, 3), ("rec aux", 13, 16, 3), ("keccak", 14, 2432, 1), ("sweep14", 14, 2048, 1),
          ("sweep13", 13, 2048, 1), ("cpu", 12, 192, 1), ("arith", 16, 128, 1)]
for name, log_n, C, r in shapes:
    n = 1 << log_n
    v = torch.randint(0, 2**62, (C, n), dtype=torch.int64, device="cuda")
    o = torch.empty_like(v)
    coeffs, _ = bpg.ops.lde_batch(v, r)
    row = "%-10s 2^%d x %4d r=%d |" % (name, log_n, C, r)
    for mode in (1, 2, 0):
        L.bp_tune_ntt_split(mode)
        t_i = timeit(lambda: bpg.ops.intt_batch(v, o))
        t_l = timeit(lambda: bpg.ops.lde_batch(coeffs, r, from_coeffs=True))
        row += " mode %d: intt %.3f ms %5.0f GB/s, lde %.3f ms %5.0f GB/s |" % (
            mode, t_i, 16 * n * C / t_i / 1e6, t_l, 8 * n * C * (1 + (1 << r)) / t_l / 1e6)
    L.bp_tune_ntt_split(0)
    t_ip = timeit(lambda: bpg.ops.ntt_batch_(v, bpg.ops.NTT_INV_NAT2BR))
    print(row + " in place intt %.3f ms %5.0f GB/s" % (t_ip, 16 * n * C / t_ip / 1e6), flush=True)
