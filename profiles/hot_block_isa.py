#!/usr/bin/env python3
"""The bounce block(s) of a kernel from ascendpathtracing_amd/csrc/render_kernels.s (`make -C ascendpathtracing_amd/csrc asm`):
full ISA of the largest packed-math basic block plus the per-class instruction histogram of every block that holds more
than 50 v_pk_* instructions.   python profiles/hot_block_isa.py <mangled-name substring> > profiles/<tag>_hot_block_isa.txt"""
import re, sys
from collections import Counter
key = sys.argv[1]
s = open(sys.argv[2] if len(sys.argv) > 2 else 'ascendpathtracing_amd/csrc/render_kernels.s').read()
m = re.search(r'^(_ZN12_GLOBAL__N_1\S*%s\S*):' % re.escape(key), s, re.M)
a = m.start(); b = s.index('.Lfunc_end', a)
blk = 'entry'; blocks = {blk: []}; order = [blk]
for l in s[a:b].split('\n'):
    t = l.strip()
    if not t or t.startswith(';'): continue
    mm = re.match(r'^(\.LBB\d+_\d+):', t)
    if mm: blk = mm.group(1); blocks[blk] = []; order.append(blk); continue
    if t.startswith('.') or t.endswith(':'): continue
    blocks[blk].append(t.split(';')[0].rstrip())
def cls(i):
    op = i.split()[0]
    if op == 's_nop': return 'nop'
    if op.startswith('s_'): return 'salu'
    if op.startswith(('ds_', 'global_', 'scratch_', 'buffer_', 'flat_')): return 'mem'
    if op.startswith('v_pk_'): return 'valu packed fp32 (2 flops/lane)'
    if re.match(r'v_(rsq|rcp|sqrt)_f32', op): return 'valu transcendental'
    if re.match(r'v_(cmp|cndmask|min|max|med3)', op): return 'valu compare/select/min'
    if op.startswith('v_'): return 'valu plain'
    return 'other'
print("kernel", m.group(1))
hot = [k for k in order if sum(1 for x in blocks[k] if x.startswith('v_pk_')) > 50]
for k in hot:
    c = Counter(cls(i) for i in blocks[k])
    valu = sum(v for kk, v in c.items() if kk.startswith('valu'))
    print(f"\nblock {k}: {len(blocks[k])} instructions, {valu} VALU")
    for kk, v in sorted(c.items()): print(f"    {v:4d}  {kk}")
    ops = Counter(i.split()[0] for i in blocks[k] if i.startswith('v_'))
    print("    VALU opcodes:", ", ".join(f"{o} {n}" for o, n in ops.most_common()))
big = max(hot, key=lambda k: len(blocks[k]))
print(f"\n==== ISA of block {big} ====")
for i in blocks[big]: print("    " + i)
