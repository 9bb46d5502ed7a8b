"""The NTT / LDE kernels alone on the chip, a few launches each, for rocprofv3 --pmc passes (tools/prof_round5_ntt.sh):
  * coset LDE from coefficients, 2^14 x 2432, rate 2  (ntt16_dit_kernel<14,0> / ntt_mx_dit_kernel<1>: bench `roofline_isolated`)
  * coset LDE from coefficients, 2^12 x 2048, rate 2  (ntt16_dit_kernel<12,0>)
  * inverse NTT 2^14 x 2048 and 2^12 x 2048           (ntt16_dif_kernel<14> / <12>: bench `ntt_hbm_gbps`)
With --time it prints the HIP-event time of each instead (no profiler)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proof_protocol_decoder_amd as bpg

REPS = 3


def run(time_it):
    for what, log_n, cols in (("lde", 14, 2432), ("lde", 12, 2048), ("lde", 13, 2048), ("intt", 14, 2048), ("intt", 12, 2048)):
        n = 1 << log_n
        v = torch.randint(0, 2**62, (cols, n), dtype=torch.int64, device="cuda")
        o = torch.empty_like(v)
        f = (lambda: bpg.ops.lde_batch(v, 1, from_coeffs=True)) if what == "lde" else (lambda: bpg.ops.intt_batch(v, o))
        f()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(REPS):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            f()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        alg = 8.0 * n * cols * (3 if what == "lde" else 2)
        if time_it:
            print("%-4s 2^%d x %d: %.1f us, %.0f GB/s algorithmic = %.4f of 8 TB/s" % (what, log_n, cols, best * 1e3, alg / best / 1e6, alg / best / 1e6 / 8000), flush=True)
        del v, o


if __name__ == "__main__":
    if "--persist" in sys.argv:   # the 2^14-point LDE blocks as one-shot grid / persistent prefetching workgroups
        import ctypes as C
        L = bpg.lib()
        L.bp_tune_ntt_persist.argtypes = [C.c_int, C.c_int]
        L.bp_tune_ntt_mx(0)
        for on, wgs in ((0, 256), (1, 256), (1, 512), (0, 256), (1, 256)):
            L.bp_tune_ntt_persist(on, wgs)
            print("bp_tune_ntt_persist(%d, %d)" % (on, wgs), flush=True)
            run(True)
        sys.exit(0)
    for knob in ([0, 3] if "--both" in sys.argv else [None]):
        if knob is not None:
            bpg.lib().bp_tune_ntt_mx(knob)
            print("bp_tune_ntt_mx(%d)" % knob, flush=True)
        run("--time" in sys.argv)
