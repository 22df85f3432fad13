#!/usr/bin/env python3
"""Scan ascendpathtracing_amd/csrc/render_kernels.s (`make -C ascendpathtracing_amd/csrc asm`) for the gfx950 hazard that hipcc
does not guard for inline-asm readers: a TRANS result (v_rsq / v_rcp / v_sqrt / v_exp / v_log / v_sin / v_cos, f32 or f16; the
float64 forms are not TRANS-pipe instructions but are listed too, conservatively) read by the very next instruction when that is
a non-TRANS VALU instruction.  hipcc separates such pairs of its own instructions by at least one wait state; a hit here is an
inline-asm reader scheduled right behind the producer (round 2 met one: profiles/history/r02_insitu_costs.md).  Exit code 1 on a hit."""
import re, sys
trans = re.compile(r"^v_(rsq|rcp|sqrt|exp|log|sin|cos)(_iflag|_legacy)?_(f32|f16|f64)")
vreg = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(text):
    """Every VGPR named anywhere in `text` (operand modifiers such as op_sel_hi:[1,0], clamp, mul:2, |x|, -x do not hide one)."""
    out = set()
    for m in vreg.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def scan(lines):
    """-> [(line number, kernel, instruction)]: a TRANS result read by the very next instruction when that is a non-TRANS VALU
    instruction.  A label between the two does not separate them (the reader may be reached by falling through)."""
    hits, prev, kernel = [], None, None
    for ln, line in enumerate(lines, 1):
        t = line.split(";")[0].strip()
        if not t:
            continue
        if t.endswith(":"):
            if not t.startswith("."):
                kernel, prev = t[:-1], None          # a new function: nothing falls through into it
            continue
        if t.startswith("."):
            continue
        op, _, rest = t.partition(" ")
        dst_text, _, src_text = rest.partition(",")
        if prev and op.startswith("v_") and not trans.match(op):
            if prev & regs(src_text):
                hits.append((ln, kernel, t))
        prev = regs(dst_text) if trans.match(op) else None
    return hits


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "ascendpathtracing_amd/csrc/render_kernels.s"
    hits = scan(open(path))
    for h in hits:
        print("TRANS result read by the next VALU instruction: line %d in %s: %s" % h)
    print(f"{len(hits)} hit(s) in {path}")
    sys.exit(1 if hits else 0)
