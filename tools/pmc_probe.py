"""Runs the calibration copy and the LDE kernel a few times each; used under
`rocprofv3 --pmc FETCH_SIZE` and `rocprofv3 --pmc WRITE_SIZE` (separate passes)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proof_protocol_decoder_amd as bpg
L = bpg.lib()
L.bp_debug_copy_u64.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
log_n, Cc, r = 14, 2432, 1
n = 1 << log_n
a = torch.randint(0, 2**62, (Cc, n), dtype=torch.int64, device="cuda")
b = torch.empty_like(a)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.bp_debug_copy_u64(a.data_ptr(), b.data_ptr(), a.numel(), st)   # known: 8*n*C read, 8*n*C written
torch.cuda.synchronize()
for _ in range(3):
    bpg.ops.lde_batch(a, r, from_coeffs=True)                          # algorithmic: 8*n*C*(1+2^r)
    bpg.ops.ntt_batch_(a, bpg.ops.NTT_INV_NAT2BR)                      # algorithmic: 16*n*C
torch.cuda.synchronize()
print("bytes per copy launch: read %d written %d" % (a.numel() * 8, a.numel() * 8))
