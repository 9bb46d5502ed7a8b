"""Static instruction mix of the device code of one HIP source: per kernel, the number of VALU instructions by
opcode and their issue cost by the measured classes of profiles/r2_issue_rate2.txt (simple 32-bit ALU ops 2.5
cycles per wave64, everything else 4.5).  Straight-line kernels only (the NTT kernels are fully unrolled); for
looped kernels the figure is per static instruction, not per execution.
usage: python tools/isa_mix.py proof_protocol_decoder_amd/csrc/ntt.hip [name-filter]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHEAP = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32",
         "v_add_f32", "v_not_b32", "v_lshlrev_b32"}


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    csrc = os.path.join(ROOT, "proof_protocol_decoder_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-I", csrc, "-I", os.path.join(ROOT, "include"), "-o", out, src], check=True, capture_output=True)
        txt = open(out).read()
    name, cnt = None, None
    res = []
    for line in txt.splitlines():
        m = re.match(r"(_Z\w+):\s", line)
        if m:
            name, cnt = m.group(1), collections.Counter()
            continue
        if line.startswith(".Lfunc_end") and name:
            res.append((name, cnt))
            name = None
            continue
        if name is None:
            continue
        t = line.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        cnt[t.split()[0]] += 1
    for name, cnt in res:
        if flt not in name:
            continue
        base = lambda op: re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
        valu = {k: v for k, v in cnt.items() if k.startswith("v_")}
        n = sum(valu.values())
        cheap = sum(v for k, v in valu.items() if base(k) in CHEAP)
        cyc = cheap * 2.5 + (n - cheap) * 4.5
        print("%s\n  VALU %d (cheap class %d) ~%.0f issue cycles; SALU %d, s_nop %d, LDS %d, VMEM %d, waitcnt %d" % (
            name, n, cheap, cyc, sum(v for k, v in cnt.items() if k.startswith("s_") and k not in ("s_nop", "s_waitcnt")),
            cnt["s_nop"], sum(v for k, v in cnt.items() if k.startswith("ds_")),
            sum(v for k, v in cnt.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_"))), cnt["s_waitcnt"]))
        print("   ", ", ".join("%s %d" % kv for kv in collections.Counter(valu).most_common(16)))


if __name__ == "__main__":
    main()
