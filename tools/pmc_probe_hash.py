"""One Merkle commitment of a 2^20 x 64 table (2^21 rows x 8 permutations) with the matrix-core kernels and one with one
lane per state, for `rocprofv3 --pmc ...` passes over SQ counters (how VALU and MFMA issue share a SIMD)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proof_protocol_decoder_amd as bpg
L = bpg.lib()
log_n, cols, r = 20, 64, 1
lde = torch.randint(0, 2**62, (cols, 1 << (log_n + r)), dtype=torch.int64, device="cuda")
for mx, grouped in ((1, 1), (1, 0), (0, 0)):     # matrix cores + grouped partial rounds / per round / one lane per state
    L.bp_tune_poseidon_mx(mx)
    L.bp_tune_poseidon_grouped(grouped)
    bpg.ops.merkle_commit(lde, log_n, r, 4)
    torch.cuda.synchronize()
L.bp_tune_poseidon_mx(1)
L.bp_tune_poseidon_grouped(1)
print("perms per commit:", (1 << (log_n + r)) * 9)
