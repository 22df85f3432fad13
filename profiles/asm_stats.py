#!/usr/bin/env python3
"""Instruction mix of the large basic blocks of one kernel in render_kernels.s
   python profiles/asm_stats.py <substring of mangled kernel name> [min_block_size]"""
import re, sys
from collections import Counter
s = open('ascendpathtracing_amd/csrc/render_kernels.s').read() if len(sys.argv) < 4 else open(sys.argv[3]).read()
key = sys.argv[1]; minsz = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for m in re.finditer(r'^(_Z\S+):', s, re.M):
    name = m.group(1)
    if key not in name: continue
    a = m.start(); b = s.index('.Lfunc_end', a)
    blk = 'entry'; blocks = {blk: []}; order = [blk]
    for l in s[a:b].split('\n'):
        t = l.strip()
        if not t or t.startswith(';'): continue
        mm = re.match(r'^(\.LBB\d+_\d+):', t)
        if mm: blk = mm.group(1); blocks[blk] = []; order.append(blk); continue
        if t.startswith('.') or t.endswith(':'): continue
        blocks[blk].append(t)
    print(name, 'total', sum(len(v) for v in blocks.values()))
    for k in order:
        ins = [x.split()[0] for x in blocks[k]]
        if len(ins) >= minsz:
            print(' ', k, len(ins), Counter(ins).most_common(40))
    meta = s[s.index('amdhsa.kernels'):]
    i = meta.index(name)
    chunk = meta[i:i + 1500]
    print('  ', re.findall(r'\.(sgpr_count|sgpr_spill_count|vgpr_count|vgpr_spill_count):\s+(\d+)', chunk))
