#!/usr/bin/env python3
"""Scan ascendpathtracing_amd/csrc/render_kernels.s (`make -C ascendpathtracing_amd/csrc asm`) for the gfx950 hazard that hipcc
does not guard for inline-asm readers: a TRANS result (v_rsq / v_rcp / v_sqrt / v_exp / v_log / v_sin / v_cos, f32 or f16; the
float64 forms are not TRANS-pipe instructions but are listed too, conservatively) read by the very next instruction when that is
a non-TRANS VALU instruction.  hipcc separates such pairs of its own instructions by at least one wait state; a hit here is an
inline-asm reader scheduled right behind the producer (round 2 met one: profiles/r02_insitu_costs.md).  Exit code 1 on a hit."""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "ascendpathtracing_amd/csrc/render_kernels.s"
trans = re.compile(r"^v_(rsq|rcp|sqrt|exp|log|sin|cos)(_iflag|_legacy)?_(f32|f16|f64)")
def regs(tok):
    tok = tok.strip().strip("|").lstrip("-")
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
hits, prev, kernel = [], None, None
for ln, line in enumerate(open(path), 1):
    t = line.split(";")[0].strip()
    if not t: continue
    if t.endswith(":"):
        if not t.startswith("."): kernel = t[:-1]
        prev = None
        continue
    if t.startswith("."): continue
    op, _, rest = t.partition(" ")
    ops = [x for x in rest.split(",")]
    if prev and op.startswith("v_") and not trans.match(op):
        dst = prev
        srcs = set()
        for x in ops[1:]: srcs |= regs(x.split(" ")[1] if x.startswith(" ") and " " in x.strip() and not x.strip().startswith("v") else x)
        if dst & srcs: hits.append((ln, kernel, t))
    prev = regs(ops[0]) if trans.match(op) else None
    if op.startswith("s_nop") or not (op.startswith("v_") or op.startswith("s_") or op.startswith("ds_") or op.startswith("global_") or op.startswith("scratch_")): prev = prev
for h in hits: print("TRANS result read by the next VALU instruction: line %d in %s: %s" % h)
print(f"{len(hits)} hit(s) in {path}")
sys.exit(1 if hits else 0)
