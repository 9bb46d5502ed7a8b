"""`__graft_entry__.build()` is the driver's does-it-build check: every HIP source (library, tools) must
cross-compile for gfx950 and the package must load.  CPU only (hipcc needs no GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_entry_compiles_everything():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    for rel in ("proof_protocol_decoder_amd/lib/libbpg.so", "oracle/liboracle.so", "tools/microbench", "tools/wait_probe", "tools/issue_rate"):
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
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
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
