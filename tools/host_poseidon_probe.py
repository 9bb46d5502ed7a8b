import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
import proof_protocol_decoder_amd as pkg
L = pkg.lib()
L.bp_debug_poseidon_host.argtypes = [C.c_void_p, C.c_size_t]
L.bp_tune_host_poseidon.argtypes = [C.c_int]
s = np.arange(12 * 20000, dtype=np.uint64).reshape(-1, 12)
for mode in (0, 1):
    L.bp_tune_host_poseidon(mode)
    t0 = time.perf_counter(); L.bp_debug_poseidon_host(s.ctypes.data, s.shape[0]); dt = time.perf_counter() - t0
    print("host poseidon form %d: %.2f us/perm" % (mode, dt / s.shape[0] * 1e6))
L.bp_tune_host_poseidon(0)
