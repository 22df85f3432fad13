#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (written by profiles/run_profile.sh) into the
committed evidence:  profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json and
profiles/hbm_traffic.json (what bench.py reports as roofline.traffic).
    python profiles/summarize.py <tag> [bench_json]"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
WARMUP = 3          # run_profile.sh runs bench.py --steps 20 --warmup 3
src = f"gpurun_out/prof_{tag}"
here = os.path.dirname(os.path.abspath(__file__))
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)   # gpurun merges runs into gpurun_out/: older passes may still lie there
stats = newest(f"{src}/trace/*/*_kernel_stats.csv")
shutil.copy(stats, f"{here}/{tag}_kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
main = max(rows, key=lambda r: float(r["TotalDurationNs"]))
pmc = {}
for f in sorted(newest(f"{d}/*/*_counter_collection.csv") for d in glob.glob(f"{src}/pmc[0-9]*") if os.path.isdir(d)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"] == main["Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        # (rows are in dispatch order.)  The traffic counters of the process's FIRST launch carry the write-back of the buffers the process has
        # just zero-filled (torch.zeros of the packed frame slots: 116 MB written / 42 MB fetched "by" that launch in r05, 31.1 / 0.4 by every other
        # one): the averages that feed hbm_traffic.json are over the TIMED launches, i.e. without the warm-up ones.
        timed = v[WARMUP:] if k in ("FETCH_SIZE", "WRITE_SIZE") and len(v) > WARMUP else v
        pmc[k] = {"launches": len(timed), "avg": sum(timed) / len(timed), "first_launch": v[0]}
out = {"tag": tag, "command": "python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-traffic-probe",
       "kernel": main["Name"], "calls": int(main["Calls"]), "avg_ns": float(main["AverageNs"]),
       "min_ns": float(main["MinNs"]), "max_ns": float(main["MaxNs"]), "pmc": pmc}
if "GRBM_GUI_ACTIVE" in pmc:   # summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
    out["effective_clock_ghz"] = pmc["GRBM_GUI_ACTIVE"]["avg"] / 8 / out["avg_ns"]
if "SQ_INSTS_VALU" in pmc and "SQ_WAVES" in pmc:
    out["valu_insts_per_wave"] = pmc["SQ_INSTS_VALU"]["avg"] / pmc["SQ_WAVES"]["avg"]
    out["valu_insts_per_segment"] = out["valu_insts_per_wave"] / 64.0          # 8 samples x 8 bounces per lane
    if "GRBM_GUI_ACTIVE" in pmc:
        out["simd_cycles_per_valu_inst"] = pmc["GRBM_GUI_ACTIVE"]["avg"] / 8 * 1024 / pmc["SQ_INSTS_VALU"]["avg"]
if "GRBM_GUI_ACTIVE" in pmc:
    cyc = pmc["GRBM_GUI_ACTIVE"]["avg"] / 8                      # shader cycles of the launch
    if "SQ_INSTS_VALU" in pmc:                                  # issue rate per SIMD (1024 SIMDs); the derived "VALUBusy" of
        out["valu_insts_per_simd_cycle"] = pmc["SQ_INSTS_VALU"]["avg"] / (cyc * 1024)   # older tools exceeds 100 % here, not reported
    if "SQ_THREAD_CYCLES_VALU" in pmc and "SQ_ACTIVE_INST_VALU" in pmc:   # fraction of the 64 lanes active per VALU instruction
        out["valu_lane_activity"] = pmc["SQ_THREAD_CYCLES_VALU"]["avg"] / (pmc["SQ_ACTIVE_INST_VALU"]["avg"] * 64)
    if "SQ_WAVE_CYCLES" in pmc:                                 # quad-cycles a wave is resident, summed -> mean waves per SIMD
        out["mean_waves_per_simd"] = pmc["SQ_WAVE_CYCLES"]["avg"] * 4 / (cyc * 1024)
        out["occupancy_pct_of_8_waves"] = 100.0 * out["mean_waves_per_simd"] / 8
if "SQC_ICACHE_REQ" in pmc and "SQC_ICACHE_MISSES" in pmc:   # instruction cache: does the kernel's code size (copies of the hot block per leaf shape) cost anything?
    out["icache_misses_per_request"] = (pmc["SQC_ICACHE_MISSES"]["avg"] + pmc.get("SQC_ICACHE_MISSES_DUPLICATE", {"avg": 0.0})["avg"]) / pmc["SQC_ICACHE_REQ"]["avg"]
json.dump(out, open(f"{here}/{tag}_pmc.json", "w"), indent=1)
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # rocprofv3 reports KiB; gfx950: FETCH_SIZE reads half of the bytes actually fetched (guide, HBM section)
    fetch = pmc["FETCH_SIZE"]["avg"] * 1024 * 2
    write = pmc["WRITE_SIZE"]["avg"] * 1024
    sys.path.insert(0, os.path.dirname(here))
    from ascendpathtracing_amd._lib import build_id
    json.dump({"tag": tag, "build_id": build_id(), "kernel": main["Name"], "hbm_gbps": (fetch + write) / out["avg_ns"],
               "hbm_frac_of_8TBps": (fetch + write) / out["avg_ns"] / 8000.0,
               "fetch_bytes_per_launch_corrected_x2": fetch,
               "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
               "algorithmic_bytes_per_launch": 1920 * 1080 * 15 + 512,
               "note": "separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes; FETCH_SIZE doubled per MI355X_MICROARCH.md; mean over the 20 timed launches "
                       "(the process's first launch also writes back the zero-filled buffers: see first_launch in the pmc file)"},
              open(f"{here}/hbm_traffic.json", "w"), indent=1)
if len(sys.argv) > 2:
    shutil.copy(sys.argv[2], f"{here}/{tag}_bench.json")
print(json.dumps(out, indent=1)[:1500])
