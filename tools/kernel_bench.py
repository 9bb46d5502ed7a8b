"""Times the L0 kernels at the synthetic-txn table shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proof_protocol_decoder_amd as bpg

def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best

P = 0xFFFFFFFF00000001
def rnd(shape):
    return torch.randint(0, 2**62, shape, dtype=torch.int64, device="cuda")

shapes = [("arith", 16, 128, 1), ("bytepack", 9, 128, 1), ("cpu", 12, 192, 1), ("keccak", 14, 2432, 1),
          ("ksponge", 9, 512, 1), ("logic", 12, 320, 1), ("memory", 17, 16, 1), ("recursion", 13, 135, 3),
          ("sweep12", 12, 2048, 1), ("sweep14", 14, 2048, 1), ("sweep16", 16, 256, 1), ("sweep18", 18, 256, 1), ("sweep20", 20, 64, 1)]
for name, log_n, C, r in shapes:
    n = 1 << log_n
    v = rnd((C, n))
    bpg.lib().bp_tune_ntt_mx(0)          # VALU butterflies
    t_intt_v = timeit(lambda: bpg.ops.ntt_batch_(v, bpg.ops.NTT_INV_NAT2BR))
    t_lde_v = timeit(lambda: bpg.ops.lde_batch(v, r))
    bpg.lib().bp_tune_ntt_mx(2)          # 16-point DFTs on the matrix cores
    t_intt = timeit(lambda: bpg.ops.ntt_batch_(v, bpg.ops.NTT_INV_NAT2BR))
    t_lde = timeit(lambda: bpg.ops.lde_batch(v, r))
    bpg.lib().bp_tune_ntt_mx(3)
    coeffs, lde = bpg.ops.lde_batch(v, r)
    bpg.lib().bp_tune_quad_threshold(1)  # 1: never quad (0 = automatic)
    bpg.lib().bp_tune_poseidon_mx(1)     # MDS layer on the matrix cores, 4 / 2 / 1 sets of 16 states per wave
    t_ms = []
    for sets in (4, 2, 1):
        bpg.lib().bp_tune_poseidon_mx_sets(sets)
        t_ms.append(timeit(lambda: bpg.ops.merkle_commit(lde, log_n, r, 4)))
    bpg.lib().bp_tune_poseidon_mx_sets(0)
    t_mx = t_ms[0]
    bpg.lib().bp_tune_poseidon_mx(0)     # one lane per state
    t_mk = timeit(lambda: bpg.ops.merkle_commit(lde, log_n, r, 4))
    bpg.lib().bp_tune_quad_threshold(1 << 40)   # quad-cooperative kernels (matrix-core form still off)
    t_mq = timeit(lambda: bpg.ops.merkle_commit(lde, log_n, r, 4))
    bpg.lib().bp_tune_poseidon_mx(1)
    perms = (n << r) * ((C + 7) // 8) + (n << r)
    print("%-10s logn=%2d C=%4d r=%d | intt %7.3f ms %6.0f GB/s (valu %6.0f) | intt+lde %7.3f ms %6.0f GB/s(alg) (valu %6.0f) | merkle mx4 %8.3f ms %6.3f Gperm/s mx2 %6.3f mx1 %6.3f | lane %8.3f ms %6.3f Gperm/s | quad %8.3f ms %6.3f Gperm/s" % (
        name, log_n, C, r, t_intt, 16 * n * C / t_intt / 1e6, 16 * n * C / t_intt_v / 1e6, t_lde, 8 * n * C * (2 + (1 << r)) / t_lde / 1e6, 8 * n * C * (2 + (1 << r)) / t_lde_v / 1e6, t_mx, perms / t_mx / 1e6, perms / t_ms[1] / 1e6, perms / t_ms[2] / 1e6, t_mk, perms / t_mk / 1e6, t_mq, perms / t_mq / 1e6), flush=True)
