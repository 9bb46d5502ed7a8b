"""Instruction-class mix of the kernels that issue most of a block's VALU instructions, from their gfx950 assembly
(hipcc -S, no GPU needed), and the cycle cost per wave64 VALU instruction that follows from tools/issue_rate.hip's
measurements (profiles/r2_issue_rates.txt): the cheap class -- 32-bit add / sub / logic / shift / move without a carry,
no DPP / SDWA -- issues every 2.2..2.7 cycles (2.45 used), everything with a carry, a multiplier, three operands or
64-bit operands every 4.1..4.8 (4.5 used).  A STATIC count: the kernels' hot code is straight-line, unrolled rounds /
butterflies executed the same number of times, so the static mix of a kernel's body is its dynamic mix to a few percent.
Families are weighted by their share of a txn proof's VALU instructions in the LOADED run (SQ counters,
profiles/r5_sq_loaded_by_kernel.txt).  Writes profiles/r5_valu_class_mix.txt; bench.py reads its last line.

    python tools/valu_class_mix.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "proof_protocol_decoder_amd", "csrc")
CHEAP = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_lshlrev_b32",
         "v_mov_b32", "v_not_b32", "v_bfe_u32", "v_add_f32", "v_mul_f32"}
FAMILIES = [  # (family in the SQ summary, source, mangled-name fragments of the kernel form the loaded run uses)
    ("leaf_hash_mx_kernel", "hash_kernels.hip", ["leaf_hash_mx_kernelILi4ELi3E", "leaf_hash_mx_kernelILi4E"]),
    ("ntt", "ntt.hip", ["ntt16_dit_kernelILi13ELi0E", "ntt16_dif_kernelILi13ELi0E"]),
    ("merkle_level_mx_kernel", "hash_kernels.hip", ["merkle_level_mx_kernel"]),
    ("quotient_air_kernel", "stark_kernels.hip", ["quotient_air_kernelILj8E", "quotient_air_kernelILj0E"]),
    ("pow_grind", "stark_kernels.hip", ["pow_grind_mx_kernelILi3E"]),
    ("quotient_plonk_hash_kernel", "stark_kernels.hip", ["quotient_plonk_hash_kernel"]),   # AIR 8's Poseidon-gate pass
]


def flags():
    out = subprocess.run(["make", "-s", "-C", CSRC, "print-hip-flags"], capture_output=True, text=True, check=True).stdout.split()
    return out


def kernel_bodies(src, tmp):
    s = os.path.join(tmp, src + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags() + ["-S", "--cuda-device-only", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
                    "-o", s, os.path.join(CSRC, src)], check=True, capture_output=True)
    bodies, name = {}, None
    for line in open(s):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            bodies[name] = []
            continue
        if name and line.startswith("\t.end_amdhsa_kernel"):
            name = None
        if name and re.match(r"^\t(v_|s_|ds_|global_|buffer_|scratch_)", line):
            bodies[name].append(line.split()[0])
    return bodies


def main():
    shares = {}
    sq = next(f for f in ("r5_sq_loaded_by_kernel.txt", "r4_sq_loaded_by_kernel.txt") if os.path.exists(os.path.join(ROOT, "profiles", f)))
    txt = open(os.path.join(ROOT, "profiles", sq)).read()
    for m in re.finditer(r"^(\S*)\s+VALU ([0-9.e+]+)", txt, re.M):
        shares[m.group(1)] = float(m.group(2))
    total = sum(shares.values())
    rows, acc_w, acc_c = [], 0.0, 0.0
    with tempfile.TemporaryDirectory() as tmp:
        cache = {}
        for fam, src, frags in FAMILIES:
            if src not in cache:
                cache[src] = kernel_bodies(src, tmp)
            n_valu = n_cheap = n_mfma = 0
            used = []
            for frag in frags:
                for name, ops in cache[src].items():
                    if frag in name:
                        used.append(name)
                        for op in ops:
                            base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
                            if op.startswith("v_mfma"):
                                n_mfma += 1
                            elif op.startswith("v_"):
                                n_valu += 1
                                if base in CHEAP and not op.endswith(("_dpp", "_sdwa")):
                                    n_cheap += 1
                if used:
                    break
            if not n_valu:
                continue
            f_cheap = n_cheap / n_valu
            cyc = 2.45 * f_cheap + 4.5 * (1 - f_cheap)
            share = shares.get(fam, 0.0) / total
            rows.append((fam, used[0][:60], n_valu, n_mfma, f_cheap, cyc, share))
            acc_w += share
            acc_c += share * cyc
    head = next((l for l in open(os.path.join(ROOT, ".head_for_profiles"))), "?").strip() if os.path.exists(os.path.join(ROOT, ".head_for_profiles")) else "?"
    with open(os.path.join(ROOT, "profiles", "r5_valu_class_mix.txt"), "w") as out:
        out.write("# python tools/valu_class_mix.py at HEAD %s: static VALU class mix of the kernel forms the loaded block run uses\n" % head)
        out.write("# cheap class (2.45 cycles): %s; everything else 4.5 cycles (profiles/r2_issue_rates.txt)\n" % ", ".join(sorted(CHEAP)))
        out.write("%-26s %-62s %8s %6s %7s %7s %7s\n" % ("family", "kernel", "VALU", "MFMA", "cheap", "cycles", "share"))
        for r in rows:
            out.write("%-26s %-62s %8d %6d %7.3f %7.2f %7.3f\n" % r)
        rest = 1 - acc_w
        out.write("families above: %.3f of a txn proof's VALU instructions (SQ counters of the loaded run); the rest (%.3f) is priced at 4.5\n" % (acc_w, rest))
        out.write("weighted cycles per VALU instruction (block mix): %.2f\n" % (acc_c + rest * 4.5))
    print(open(os.path.join(ROOT, "profiles", "r5_valu_class_mix.txt")).read())


if __name__ == "__main__":
    main()
