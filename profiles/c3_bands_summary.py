#!/usr/bin/env python3
"""Per band size (= per rank count of the split) of profiles/c3_bands_profile.sh's passes: launches, mean kernel duration, HBM bytes and GB/s,
VALU instructions per SIMD-cycle, waves per SIMD, lane activity (bench.py's formulas).  python3 profiles/c3_bands_summary.py <out-dir>"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

out = sys.argv[1]
KEY = "render_frame_kernel<0, 0, 8, false, true>"
NPIX = 4096 * 4096
by = collections.defaultdict(lambda: collections.defaultdict(list))          # grid size -> counter -> values
dur = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*_counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if KEY in r["Kernel_Name"]:
            by[int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob(os.path.join(out, "trace", "**", "*_kernel_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if KEY in r["Kernel_Name"]:
            dur[int(r["Grid_Size_X"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
rows = []
for grid in sorted(by, reverse=True):
    avg = {c: sum(v) / len(v) for c, v in by[grid].items()}
    ns = sum(dur[grid]) / max(1, len(dur[grid])) if dur.get(grid) else None
    pixels = grid // 256 * 8                                             # 8 pixels per workgroup of 256 threads
    ranks = round(NPIX / pixels)
    row = {"ranks_of_the_split": ranks, "band_pixels": pixels, "launches": len(dur.get(grid, [])), "kernel_ms": round(ns / 1e6, 3) if ns else None}
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg and ns:
        traffic = 2.0 * avg["FETCH_SIZE"] * 1024 + avg["WRITE_SIZE"] * 1024
        row.update({"hbm_bytes_per_launch": round(traffic), "algorithmic_bytes": pixels * 15 + 512, "hbm_gbps": round(traffic / ns, 3)})
    if all(c in avg for c in ("GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES")) and ns:
        row.update(bench.derive_valu(avg, ns, pixels * 4 * 256 * 8))
    rows.append(row)
    print(json.dumps(row))
json.dump(rows, open(os.path.join(out, "summary.json"), "w"), indent=1)
