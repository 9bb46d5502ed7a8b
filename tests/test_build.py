"""`__graft_entry__.build()` is the driver's does-it-build check: every HIP source (library, tools) must
cross-compile for gfx950 and the package must load.  CPU only (hipcc needs no GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the code-generation flags of csrc/Makefile
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def test_build_entry_compiles_everything():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    for rel in ("proof_protocol_decoder_amd/lib/libbpg.so", "oracle/liboracle.so", "tools/microbench", "tools/wait_probe", "tools/issue_rate",
                "tools/mfma_probe"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel


def test_no_kernel_spills_sgprs(tmp_path):
    """The carry-chain arithmetic (csrc/gl.hpp) keeps carries as wave masks in SGPR pairs written by inline asm.
    If register pressure made the compiler park such a mask in a VGPR lane (v_writelane) right after the asm
    wrote it, the read would need 2 wait states that the hazard recogniser cannot provide (it does not see writes
    inside inline asm).  The kernels are therefore kept free of SGPR spills; this compiles them and checks."""
    import re
    import subprocess
    csrc = os.path.join(ROOT, "proof_protocol_decoder_amd", "csrc")
    bad = []
    for src in ("hash_kernels.hip", "ntt.hip", "stark_kernels.hip"):
        out = tmp_path / (src + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + HIPCC_FLAGS + ["-S", "--cuda-device-only",
                        "-I", csrc, "-I", os.path.join(ROOT, "include"), "-o", str(out), os.path.join(csrc, src)],
                       check=True, capture_output=True)
        name = None
        for line in out.read_text().splitlines():
            m = re.match(r"\s+\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
            m = re.match(r"\s+\.sgpr_spill_count:\s+(\d+)", line)
            if m and int(m.group(1)):
                bad.append((src, name, int(m.group(1))))
    assert not bad, bad


def _vregs(tok):
    """register numbers named by one operand token: v7 -> {7}, v[4:5] -> {4, 5}"""
    import re
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def test_no_dpp_reads_a_fresh_asm_result(tmp_path):
    """A DPP / permlane read of a VGPR needs 2 wait states after the VALU write of that register.  The
    compiler's hazard recogniser inserts them for its own instructions but cannot see writes made inside
    inline asm (csrc/poseidon.cuh pins an `s_nop 1` by hand).  Scan the device assembly: no DPP instruction
    may read a VGPR written inside an ASMSTART/ASMEND region fewer than 2 wait states earlier."""
    import re
    import subprocess
    csrc = os.path.join(ROOT, "proof_protocol_decoder_amd", "csrc")
    bad = []
    for src in ("hash_kernels.hip", "stark_kernels.hip"):
        out = tmp_path / (src + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + HIPCC_FLAGS + ["-S", "--cuda-device-only",
                        "-I", csrc, "-I", os.path.join(ROOT, "include"), "-o", str(out), os.path.join(csrc, src)],
                       check=True, capture_output=True)
        in_asm = False
        recent = []  # (wait states since, regs written inside asm)
        n_dpp = 0
        for ln, line in enumerate(out.read_text().splitlines(), 1):
            t = line.strip()
            if t.startswith(";ASMSTART") or t.startswith("; ASMSTART") or "ASMSTART" in t and t.startswith(";"):
                in_asm = True
                continue
            if "ASMEND" in t and t.startswith(";"):
                in_asm = False
                continue
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                continue
            op = t.split()[0]
            if not re.match(r"[sv]_|ds_|global_|buffer_|flat_|scratch_", op):
                continue
            operands = [x.strip() for x in t[len(op):].split(";")[0].replace(" quad_perm", ", quad_perm").split(",")]
            states = 1
            if op == "s_nop":
                states = int(operands[0], 0) + 1
            is_dpp = "_dpp" in op or "quad_perm" in t or "row_" in t or op.startswith("v_permlane")
            is_swap = op.startswith("v_permlane") and "swap" in op  # reads AND writes both of its operands
            if is_dpp:
                n_dpp += 1
                srcs = set()
                for o in (operands if is_swap else operands[1:]):
                    srcs |= _vregs(o.split()[0] if o else o)
                for age, regs in recent:
                    if age < 2 and regs & srcs:
                        bad.append((src, ln, t))
            recent = [(age + states, regs) for age, regs in recent if age + states < 2]
            if in_asm and op.startswith("v_") and operands:
                recent.append((0, _vregs(operands[0]) | (_vregs(operands[1]) if is_swap and len(operands) > 1 else set())))
        assert n_dpp > 0 or src != "hash_kernels.hip", "scanner found no DPP instruction in " + src
    assert not bad, bad
