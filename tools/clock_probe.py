"""Kernel times right after a few seconds of saturating load (clock / power droop): the LDE and the Merkle
commit are ~12 % slower for several seconds, which is why bench.py takes its single-stream roofline leg first."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import proof_protocol_decoder_amd as bpg
def t_lde(v):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); bpg.ops.lde_batch(v, 1); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)
def t_hash(lde):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); bpg.ops.merkle_commit(lde, 13, 3, 4); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)
v = torch.randint(0, 2**62, (2432, 1 << 14), dtype=torch.int64, device="cuda")
w = torch.randint(0, 2**62, (135, 1 << 16), dtype=torch.int64, device="cuda")
for i in range(3): t_lde(v); t_hash(w)
print("cold: lde %.3f ms hash %.3f ms" % (t_lde(v), t_hash(w)))
# heavy load for ~6 s: big Merkle commits
big = torch.randint(0, 2**62, (64, 1 << 20), dtype=torch.int64, device="cuda")
t0 = time.time()
while time.time() - t0 < 6: bpg.ops.merkle_commit(big, 19, 1, 4)
torch.cuda.synchronize()
for k in range(12):
    print("t=%.1f s after load: lde %.3f ms hash %.3f ms" % (time.time() - t0 - 6, t_lde(v), t_hash(w)), flush=True)
    time.sleep(0.4)
