#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel of the library, from hipcc's kernel-resource-usage remarks:
   python profiles/kernel_resources.py            (runs `make asm` in ascendpathtracing_amd/csrc)"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-B", "-C", os.path.join(root, "ascendpathtracing_amd", "csrc"), "asm"], capture_output=True, text=True).stderr
rows, cur = [], None
for l in out.split("\n"):
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", l)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": subprocess.run(["c++filt", t.split(": ", 1)[1]], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print(f"{'SGPR':>5}{'VGPR':>5}{'scratch':>8}{'waves':>6}{'sSpill':>7}{'vSpill':>7}{'LDS':>7}  kernel")
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["name"])
    n = re.sub(r"\(.*", "", n)
    print(f"{r.get('TotalSGPRs','?'):>5}{r.get('VGPRs','?'):>5}{r.get('ScratchSize [bytes/lane]','?'):>8}{r.get('Occupancy [waves/SIMD]','?'):>6}"
          f"{r.get('SGPRs Spill','?'):>7}{r.get('VGPRs Spill','?'):>7}{r.get('LDS Size [bytes/block]','?'):>7}  {n}")
