#!/usr/bin/env python3
"""Instruction mix of every loop body (backward branch target .. branch) of one kernel in a gfx950 .s file.
usage: loop_isa_stats.py file.s kernel-name-substring"""
import re
import sys

s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z[\w]+):\s*; @.*?\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    if sys.argv[2] not in m.group(1):
        continue
    body = m.group(2).split('\n')
    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = i
    print(m.group(1))
    for i, l in enumerate(body):
        mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            code = [x for x in body[labels[mm.group(1)]:i + 1] if re.match(r'^\s+[a-z]', x)]
            c = lambda pat: sum(1 for x in code if re.match(r'^\s+' + pat, x))
            nopc = sum(int(re.match(r'^\s+s_nop (\d+)', x).group(1)) + 1 for x in code if re.match(r'^\s+s_nop', x))
            print("  %s: %d lines | VALU %d (mad_u64 %d, cheap %d) MFMA %d s_nop %d (%d states) permlane %d ds %d accvgpr %d salu %d" % (
                mm.group(1), len(code), c(r'v_(?!mfma)'), c('v_mad_u64'),
                c(r'v_(mov_b32|add_u32|sub_u32|xor_b32|and_b32|or_b32|lshlrev_b32|lshrrev_b32)'), c('v_mfma'),
                c('s_nop'), nopc, c('v_permlane'), c('ds_'), c('v_accvgpr'), c(r's_(?!nop|waitcnt)')))
