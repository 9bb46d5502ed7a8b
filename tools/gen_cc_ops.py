#!/usr/bin/env python3
"""Emits proof_protocol_decoder_amd/csrc/gl_cc.inc: N-wide (N = 3, 4) carry-chain instruction groups
for the Goldilocks modular multiply (gl.hpp).  One asm statement holds the same instruction for N
independent elements, so the consumer of a carry mask is always N-1 >= 2 instructions behind its
producer: the 2 wait states gfx940+ needs between a VALU SGPR write and a VALU read of that SGPR are
met by construction, without s_nop."""
import os

OPS = {
    # name: (asm line, in/out operand (tied: the result overwrites it), carry-out?, inputs)
    # every VGPR result is written IN PLACE over its first operand: no early-clobber temporaries, so a
    # group of four multiplies needs ~11 live VGPRs per element instead of ~20.
    "mad_co":     ("v_mad_u64_u32 {x}, {co}, {a}, {b}, {x}", ("uint64_t", "x"), True, [("uint32_t", "v", "a"), ("uint32_t", "v", "b")]),      # x += a*b
    "mad_eps_co": ("v_mad_u64_u32 {x}, {co}, {a}, -1, {x}", ("uint64_t", "x"), True, [("uint32_t", "v", "a")]),                               # x += a*(2^32-1)
    "add_co":     ("v_add_co_u32_e64 {x}, {co}, {x}, {b}", ("uint32_t", "x"), True, [("uint32_t", "v", "b")]),                                # x += b
    "addc_co":    ("v_addc_co_u32_e64 {x}, {co}, {x}, {b}, {ci}", ("uint32_t", "x"), True, [("uint32_t", "v", "b"), ("mask", "s", "ci")]),    # x += b + ci
    "addc0_co":   ("v_addc_co_u32_e64 {x}, {co}, {x}, 0, {ci}", ("uint32_t", "x"), True, [("mask", "s", "ci")]),                              # x += ci
    "sub_co":     ("v_sub_co_u32_e64 {x}, {co}, {x}, {b}", ("uint32_t", "x"), True, [("uint32_t", "v", "b")]),                                # x -= b
    "subb_co":    ("v_subb_co_u32_e64 {x}, {co}, {x}, {b}, {ci}", ("uint32_t", "x"), True, [("uint32_t", "v", "b"), ("mask", "s", "ci")]),    # x -= b + ci
    "subb0_co":   ("v_subbrev_co_u32_e64 {x}, {co}, 0, {x}, {ci}", ("uint32_t", "x"), True, [("mask", "s", "ci")]),                           # x -= ci
    "add_m1_co":  ("v_add_co_u32_e64 {x}, {co}, {x}, -1", ("uint32_t", "x"), True, []),                                                          # x += 2^32-1
    "sel":        ("v_cndmask_b32_e64 {x}, {x}, {b}, {ci}", ("uint32_t", "x"), False, [("uint32_t", "v", "b"), ("mask", "s", "ci")]),             # x = ci ? b : x
    "sel_eps":    ("v_cndmask_b32_e64 {x}, 0, -1, {ci}", ("uint32_t", "x"), False, [("mask", "s", "ci")]),                                    # x = ci ? 2^32-1 : 0
    # carry-out not needed: it goes to VCC (clobbered), which frees an SGPR pair per element -- the Poseidon kernels
    # keep 24 round-constant SGPRs live and must not spill masks (tests/test_build.py)
    "addc0_cv":   ("v_addc_co_u32_e64 {x}, vcc, {x}, 0, {ci}", ("uint32_t", "x"), False, [("mask", "s", "ci")]),                              # x += ci
    "addc_cv":    ("v_addc_co_u32_e64 {x}, vcc, {x}, {b}, {ci}", ("uint32_t", "x"), False, [("uint32_t", "v", "b"), ("mask", "s", "ci")]),    # x += b + ci
    "subb0_cv":   ("v_subbrev_co_u32_e64 {x}, vcc, 0, {x}, {ci}", ("uint32_t", "x"), False, [("mask", "s", "ci")]),                           # x -= ci
    "mad_eps_cv": ("v_mad_u64_u32 {x}, vcc, {a}, -1, {x}", ("uint64_t", "x"), False, [("uint32_t", "v", "a")]),                               # x += a*(2^32-1), no carry wanted
    "sel_one":    ("v_cndmask_b32_e64 {x}, {x}, 1, {ci}", ("uint32_t", "x"), False, [("mask", "s", "ci")]),                                   # x = ci ? 1 : x
    # out-of-place forms (x is write-only): an operand that is still needed afterwards costs no v_mov copy
    "add_co_o":   ("v_add_co_u32_e64 {x}, {co}, {a}, {b}", ("uint32_t", "x"), True, [("uint32_t", "v", "a"), ("uint32_t", "v", "b")]),          # x = a + b
    "addc_co_o":  ("v_addc_co_u32_e64 {x}, {co}, {a}, {b}, {ci}", ("uint32_t", "x"), True, [("uint32_t", "v", "a"), ("uint32_t", "v", "b"), ("mask", "s", "ci")]),
    "sub_co_o":   ("v_sub_co_u32_e64 {x}, {co}, {a}, {b}", ("uint32_t", "x"), True, [("uint32_t", "v", "a"), ("uint32_t", "v", "b")]),          # x = a - b
    "subb_co_o":  ("v_subb_co_u32_e64 {x}, {co}, {a}, {b}, {ci}", ("uint32_t", "x"), True, [("uint32_t", "v", "a"), ("uint32_t", "v", "b"), ("mask", "s", "ci")]),
    "add_m1_co_o": ("v_add_co_u32_e64 {x}, {co}, {a}, -1", ("uint32_t", "x"), True, [("uint32_t", "v", "a")]),                                   # x = a + 2^32-1
    "addc0_co_o": ("v_addc_co_u32_e64 {x}, {co}, {a}, 0, {ci}", ("uint32_t", "x"), True, [("uint32_t", "v", "a"), ("mask", "s", "ci")]),          # x = a + ci
}


def emit(n):
    out = []
    for name, (line, (xt, xa), has_co, ins) in OPS.items():
        write_only = name == "sel_eps" or name.endswith("_o")
        clobber = ' : "vcc"' if name.endswith("_cv") else ""
        params = [f"{xt} (&{xa})[{n}]"] + ([f"mask (&co)[{n}]"] if has_co else []) + [f"const {t} (&{a})[{n}]" for t, _, a in ins]
        out.append(f"__device__ __forceinline__ void {name}(" + ", ".join(params) + ") {")
        idx, slots = 0, {}
        names = [xa] + (["co"] if has_co else []) + [a for _, _, a in ins]
        for a in names:
            for i in range(n):
                slots[(a, i)] = idx
                idx += 1
        text = "\\n\\t".join(line.format(**{a: f"%{slots[(a, i)]}" for a in names}) for i in range(n))
        # an output that is not also an input may not share a register with a later element's inputs
        o = [f'"{"=&v" if write_only else "+v"}"({xa}[{i}])' for i in range(n)]
        if has_co:
            o += [f'"=&s"(co[{i}])' for i in range(n)]  # a carry-out must not land on a later element's carry-in
        ii = ", ".join(f'"{c}"({a}[{i}])' for _, c, a in ins for i in range(n))
        out.append(f'  asm("{text}"\n      : {", ".join(o)}\n      : {ii}{clobber});')
        out.append("}")
    return "\n".join(out)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "..", "proof_protocol_decoder_amd", "csrc", "gl_cc.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_cc_ops.py -- do not edit.  N-wide carry-chain instruction groups.\n")
        for n in (3, 4):
            f.write(f"// ---- N = {n}\n" + emit(n) + "\n")
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
