#!/bin/bash
# PMC passes (rocprofv3, counters only + kernel trace) for one case of profiles/debug/queue_run.py.
#   bash profiles/pmc_case.sh <tag> <case> [kernel-name substring]     -> gpurun_out/prof_<tag>/pmc.json
set -u
TAG=${1:-q8}
CASE=${2:-c2_retire}
KEY=${3:-queue8}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
CMD="python3 profiles/debug/queue_run.py --case $CASE --reps 2"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i ($grp) failed"
done
python3 - "$OUT" "$KEY" "$CASE" <<'PY'
import collections, csv, glob, json, sys
out, key, case = sys.argv[1:4]
agg = collections.defaultdict(list)
dur = []
for f in sorted(glob.glob(out + "/pmc*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob(out + "/pmc*/*/*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
res = {"case": case, "kernel_key": key, "launches": len(dur), "avg_ns": sum(dur) / max(1, len(dur))}
res.update({k: sum(v) / len(v) for k, v in agg.items()})
if "GRBM_GUI_ACTIVE" in res and "SQ_INSTS_VALU" in res:
    cyc = res["GRBM_GUI_ACTIVE"] / 8
    res["simd_cycles_per_valu_inst"] = cyc * 1024 / res["SQ_INSTS_VALU"]
    res["valu_insts_per_wave"] = res["SQ_INSTS_VALU"] / res["SQ_WAVES"]
    res["mean_waves_per_simd"] = res["SQ_WAVE_CYCLES"] * 4 / (cyc * 1024)
    if "SQ_THREAD_CYCLES_VALU" in res:
        res["valu_lane_activity"] = res["SQ_THREAD_CYCLES_VALU"] / (res["SQ_ACTIVE_INST_VALU"] * 64)
json.dump(res, open(out + "/pmc.json", "w"), indent=1)
print(json.dumps(res))
PY
