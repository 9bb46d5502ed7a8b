"""One Merkle commitment of a 2^20 x 64 table (2^21 rows x 8 permutations) with the matrix-core kernels and one with one
lane per state, for `rocprofv3 --pmc ...` passes over SQ counters (how VALU and MFMA issue share a SIMD)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proof_protocol_decoder_amd as bpg
L = bpg.lib()
log_n, cols, r = 20, 64, 1
lde = torch.randint(0, 2**62, (cols, 1 << (log_n + r)), dtype=torch.int64, device="cuda")
# matrix cores + three groups (all 22 partial rounds) / two groups (rounds 4..19) / per round; one lane per state
for mx, grouped in ((1, 3), (1, 2), (1, 0), (0, 0)):
    L.bp_tune_poseidon_mx(mx)
    L.bp_tune_poseidon_grouped(grouped)
    bpg.ops.merkle_commit(lde, log_n, r, 4)
    torch.cuda.synchronize()
L.bp_tune_poseidon_mx(1)
L.bp_tune_poseidon_grouped(3)
print("perms per commit:", (1 << (log_n + r)) * 9)
