"""`__graft_entry__.build()` is the driver's does-it-build check: every HIP source (library, tools) must
cross-compile for gfx950 and the package must load.  CPU only (hipcc needs no GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import subprocess

# the code-generation flags csrc/Makefile gives the kernel sources (it probes the optional MFMA form flag)
HIPCC_FLAGS = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "proof_protocol_decoder_amd", "csrc"),
                              "print-hip-flags"], check=True, capture_output=True, text=True).stdout.split()


_ASM_CACHE = {}


def device_assembly(src, tmp_path):
    """The gfx950 assembly of csrc/<src> (hipcc -S, device only), compiled once per test session: three scanners read it."""
    if src not in _ASM_CACHE:
        csrc = os.path.join(ROOT, "proof_protocol_decoder_amd", "csrc")
        out = tmp_path / (src + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + HIPCC_FLAGS + ["-S", "--cuda-device-only",
                        "-I", csrc, "-I", os.path.join(ROOT, "include"), "-o", str(out), os.path.join(csrc, src)],
                       check=True, capture_output=True)
        _ASM_CACHE[src] = out.read_text()
    return _ASM_CACHE[src]


def test_build_entry_compiles_everything():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    for rel in ("proof_protocol_decoder_amd/lib/libbpg.so", "oracle/liboracle.so", "tools/microbench", "tools/wait_probe", "tools/issue_rate",
                "tools/mfma_probe"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel


def test_no_throughput_kernel_spills_sgprs(tmp_path):
    """The carry-chain arithmetic (csrc/gl.hpp) keeps carries as wave masks in SGPR pairs written by inline asm.
    If register pressure made the compiler park such a mask in a VGPR lane (v_writelane) right after the asm
    wrote it, the read would need 2 wait states that the hazard recogniser cannot provide (it does not see writes
    inside inline asm).  The rule itself is checked instruction by instruction on every kernel by
    test_no_valu_reads_a_fresh_asm_carry_mask (which also sees v_writelane / v_readlane); on top of it the
    throughput kernels (Poseidon, NTT) are kept free of SGPR spills altogether, which is also what their speed
    wants.  The AIR-generic quotient kernel of stark_kernels.hip holds ~50 SGPRs of kernel arguments next to the
    masks and spills a handful of those arguments (never a mask: the scan proves it), so it is exempt here, and so is
    AIR 8's Poseidon-gate pass (the same arguments next to 24 round-constant SGPRs: 4 spills)."""
    import re
    csrc = os.path.join(ROOT, "proof_protocol_decoder_amd", "csrc")
    bad, scratch = [], []
    for src in ("hash_kernels.hip", "ntt.hip", "stark_kernels.hip"):
        asm_text = device_assembly(src, tmp_path)
        name = None
        for line in asm_text.splitlines():
            m = re.match(r"\s+\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
            m = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
            if m and int(m.group(1)):
                # scratch (VGPR spills or per-lane arrays) is a decision, not an accident (HISTORY.md section 10): the
                # kernels that have some today -- the two Keccak witness generators (per-lane 25-lane states), the
                # Keccak-f evaluator held to three waves per SIMD, two opt-in NTT forms -- are listed with a cap; a new
                # one, or one that grows, fails the build check
                known = {"quotient_air_kernelILj1E": 128, "ntt16_dit_persist_kernel": 160, "keccak_trace_kernel": 512,
                         "keccak_sponge_trace_kernel": 384, "leaf_hash_rows_kernel": 16, "ntt_mx_dit_kernelILi2E": 64}
                cap = next((v for k, v in known.items() if k in name), 0)
                if int(m.group(1)) > cap:
                    scratch.append((src, name, int(m.group(1)), "allowed %d" % cap))
            m = re.match(r"\s+\.sgpr_spill_count:\s+(\d+)", line)
            if m and int(m.group(1)) and "quotient_air_kernel" not in name and "quotient_plonk_hash_kernel" not in name:
                bad.append((src, name, int(m.group(1))))
            m = re.match(r"\s+\.sgpr_spill_count:\s+(\d+)", line)
            if m and ("quotient_air_kernel" in name or "quotient_plonk_hash_kernel" in name):
                # 4 .. 15 for most AIRs, 22 for the Keccak sponge (2414 columns, three nested rolled loops)
                assert int(m.group(1)) <= 40, (name, "spills far more than kernel arguments and wave-uniform column offsets: look at it")
    assert not bad, bad
    assert not scratch, ("kernels with scratch", scratch)


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
        asm_text = device_assembly(src, tmp_path)
        in_asm = False
        recent = []  # (wait states since, regs written inside asm)
        n_dpp = 0
        for ln, line in enumerate(asm_text.splitlines(), 1):
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


def _sregs(tok):
    """scalar registers named by one operand token: s7 -> {7}, s[4:5] -> {4, 5}, vcc -> {"vcc"}"""
    import re
    tok = tok.split()[0] if tok.split() else tok
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return {"vcc"}
    return set()


def scan_carry_mask_hazards(text):
    """-> (violations, number of VALU writes of scalar registers inside asm regions) for one device assembly text."""
    import re
    bad, n_asm_writes = [], 0
    in_asm = False
    recent = []  # (wait states since the write, scalar registers a VALU instruction wrote, was it inside asm)
    for ln, line in enumerate(text.splitlines(), 1):
        t = line.strip()
        if "ASMSTART" in t and t.startswith(";"):
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
        operands = [x.strip() for x in t[len(op):].split(";")[0].split(",")]
        states = int(operands[0], 0) + 1 if op == "s_nop" else 1
        n_dst = 2 if ("_co_" in op or op.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale"))) else 1
        if op.startswith("v_"):
            reads = set()
            for o in operands[n_dst:]:
                reads |= _sregs(o)
            # the compiler protects pairs of its own instructions; everything with one end inside asm is ours
            if any(age < 2 and regs & reads and (w_in_asm or in_asm) for age, regs, w_in_asm in recent):
                bad.append((ln, t))
        if op.startswith("s_") and operands and not op.startswith(("s_nop", "s_cmp", "s_waitcnt", "s_cbranch")):
            # a scalar-unit write replaces the mask: the later read is of THAT value (no VALU-write hazard)
            over = _sregs(operands[0])
            recent = [(age, regs - over, a) for age, regs, a in recent]
        recent = [(age + states, regs, a) for age, regs, a in recent if age + states < 2 and regs]
        if op.startswith("v_"):
            written = set()
            for o in operands[:n_dst]:
                written |= _sregs(o)
            if written:
                n_asm_writes += in_asm
                recent.append((0, written, in_asm))
    return bad, n_asm_writes


def test_carry_mask_scanner_sees_a_violation():
    ok = """
	;;#ASMSTART
	v_add_co_u32_e64 v3, s[8:9], v3, v4
	v_add_co_u32_e64 v5, s[10:11], v5, v6
	v_add_co_u32_e64 v7, s[12:13], v7, v8
	v_addc_co_u32_e64 v4, s[8:9], v5, v1, s[8:9]
	;;#ASMEND
	;;#ASMSTART
	v_add_co_u32_e64 v3, s[8:9], v3, v4
	;;#ASMEND
	;;#ASMSTART
	s_nop 1
	v_addc_co_u32_e64 v4, s[8:9], v5, v1, s[8:9]
	;;#ASMEND
	;;#ASMSTART
	v_add_co_u32_e64 v3, s[6:7], v3, v4
	;;#ASMEND
	s_mov_b64 s[6:7], 0xffffffff
	v_lshl_add_u64 v[6:7], v[2:3], 0, s[6:7]
"""
    assert scan_carry_mask_hazards(ok) == ([], 7)
    for between in ("", "\tv_mov_b32 v9, v10\n", "\ts_nop 0\n"):
        bad_text = ("\t;;#ASMSTART\n\tv_add_co_u32_e64 v3, s[8:9], v3, v4\n\t;;#ASMEND\n" + between +
                    "\tv_cndmask_b32_e64 v1, v2, v3, s[8:9]\n")
        bad, n = scan_carry_mask_hazards(bad_text)
        assert n == 1 and len(bad) == 1, (between, bad)
    bad, _ = scan_carry_mask_hazards("\t;;#ASMSTART\n\tv_cmp_lt_u64_e64 vcc, v[1:2], v[3:4]\n\t;;#ASMEND\n"
                                     "\tv_cndmask_b32_e32 v1, v2, v3, vcc\n")
    assert len(bad) == 1
    # a spilled mask coming back (v_readlane, the compiler's own instruction) read by asm right away: ours to catch;
    # the same pair outside asm is the hazard recogniser's business
    reload = "\tv_readlane_b32 s8, v116, 3\n\tv_readlane_b32 s9, v116, 4\n"
    bad, _ = scan_carry_mask_hazards(reload + "\t;;#ASMSTART\n\tv_addc_co_u32_e64 v4, s[10:11], v5, v1, s[8:9]\n\t;;#ASMEND\n")
    assert len(bad) == 1
    bad, _ = scan_carry_mask_hazards(reload + "\tv_addc_co_u32_e64 v4, s[10:11], v5, v1, s[8:9]\n")
    assert bad == []
    # spilling a fresh mask: v_writelane is a VALU read of the SGPR
    bad, _ = scan_carry_mask_hazards("\t;;#ASMSTART\n\tv_add_co_u32_e64 v3, s[8:9], v3, v4\n\t;;#ASMEND\n"
                                     "\tv_writelane_b32 v116, s8, 0\n")
    assert len(bad) == 1


def test_no_valu_reads_a_fresh_asm_carry_mask(tmp_path):
    """gfx940+ needs 2 wait states between a VALU write of an SGPR (a carry-out / compare mask) and a VALU read of
    it.  The hazard recogniser inserts them for the compiler's own instructions but does not look inside inline
    asm, and the whole carry-chain arithmetic (csrc/gl.hpp, gl_cc.inc) keeps its carries in SGPR pairs written
    there: tools/gen_cc_ops.py interleaves 3-4 independent elements per asm statement so that every consumer is
    >= 2 instructions behind its producer, and the one-element forms carry `s_nop 1`.  This scans the device
    assembly for the rule itself: no VALU instruction (inside asm or emitted by the compiler) may read an SGPR
    that a VALU instruction inside an ASMSTART/ASMEND region wrote fewer than 2 wait states earlier, and no VALU
    instruction inside such a region may read one that ANY VALU instruction wrote that recently (a v_readlane that
    brings a spilled mask back, a compiler-made compare)."""
    csrc = os.path.join(ROOT, "proof_protocol_decoder_amd", "csrc")
    bad, n_asm_writes = [], 0
    for src in ("hash_kernels.hip", "ntt.hip", "stark_kernels.hip"):
        asm_text = device_assembly(src, tmp_path)
        b, n = scan_carry_mask_hazards(asm_text)
        bad += [(src,) + x for x in b]
        n_asm_writes += n
    assert n_asm_writes > 1000, "scanner found only %d carry-mask writes inside asm: is it still parsing?" % n_asm_writes
    assert not bad, bad[:10]
