"""A/B of the grouped partial rounds (bp_tune_poseidon_grouped) on the kernels that use them: Merkle commits alone on
the chip at the sizes that matter (2^21 rows x 8 permutations = bench.py's poseidon_peak; the recursion shape
2^16 x 135; the Keccak-wide S1 table 2^15 x 2432), digests compared."""
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
import proof_protocol_decoder_amd as bpg

L = bpg.lib()
L.bp_tune_poseidon_mx_sets(4)


def gperm(log_n, r, cols, reps=5):
    lde = torch.randint(0, 2**62, (cols, 1 << (log_n + r)), dtype=torch.int64, device="cuda")
    out = {}
    for on in (0, 2, 3, 0, 2, 3):
        L.bp_tune_poseidon_grouped(on)
        d = bpg.ops.merkle_commit(lde, log_n, r, 4)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            d = bpg.ops.merkle_commit(lde, log_n, r, 4)
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        perms = (1 << (log_n + r)) * ((cols + 7) // 8) + (1 << (log_n + r))
        out.setdefault(on, []).append((perms / (best * 1e-3) / 1e9, d.clone()))
    same = all((x[1] == out[0][0][1]).all().item() for v in out.values() for x in v)
    print("2^%d x %d rate %d: per-round %s Gperm/s, two groups %s, three groups %s, digests identical: %s" % (
        log_n, cols, 1 << r, ["%.3f" % x[0] for x in out[0]], ["%.3f" % x[0] for x in out[2]],
        ["%.3f" % x[0] for x in out[3]], same), flush=True)


for shape in ((20, 1, 64), (16, 3, 135), (14, 1, 2432), (17, 1, 16)):
    gperm(*shape)
