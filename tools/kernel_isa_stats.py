#!/usr/bin/env python3
"""Per-kernel resource and instruction-mix summary of a gfx950 assembly file (hipcc -S --cuda-device-only).
usage: kernel_isa_stats.py file.s [name-substring ...]"""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    want = sys.argv[2:]
    # a kernel's text runs from its label to its .end_amdhsa_kernel
    for m in re.finditer(r'^(_Z[\w]+):\s*; @.*?\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if '.amdhsa_kernel' not in body:
            continue
        if want and not any(w in name for w in want):
            continue
        code = body.split('.section')[0]
        get = lambda k: (re.search(r'\.amdhsa_%s (\d+)' % k, body) or [None, '?'])[1]
        cnt = lambda pat: len(re.findall(pat, code, re.M))
        print("%s\n   vgpr %s (accum_offset %s) sgpr %s scratch %s | VALU %d (mad_u64 %d, cheap %d) MFMA %d s_nop %d permlane %d ds %d global %d scratch_ops %d" % (
            name, get('next_free_vgpr'), get('accum_offset'), get('next_free_sgpr'), get('private_segment_fixed_size'),
            cnt(r'^\s+v_(?!mfma)'), cnt(r'^\s+v_mad_u64_u32'),
            cnt(r'^\s+v_(mov_b32|add_u32|sub_u32|xor_b32|and_b32|or_b32|lshlrev_b32|lshrrev_b32)'),
            cnt(r'^\s+v_mfma'), cnt(r'^\s+s_nop'), cnt(r'^\s+v_permlane'), cnt(r'^\s+ds_'), cnt(r'^\s+global_'),
            cnt(r'^\s+scratch_')))


if __name__ == "__main__":
    main()
