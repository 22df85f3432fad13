#!/bin/bash
# Counter passes (rocprofv3: counters + kernel trace only) of an arbitrary command, one pass per counter group.
#   bash profiles/pmc_groups.sh <tag> <kernel-name substring> "<command after -->" "<group 1>" ["<group 2>" ...]
# -> gpurun_out/prof_<tag>/pmc.json: average per launch of every counter over the launches whose kernel name contains the
# substring, plus the average duration.  A group the hardware cannot collect in one pass aborts that pass only (each pass has its own time limit).  The command must be the program itself (python3 ...), not a shell wrapper.
set -u
TAG=$1; KEY=$2; CMD=$3; shift 3
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  echo "pass $i: $grp"
  timeout -k 5 ${PMC_PASS_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i ($grp) failed"
done
python3 - "$OUT" "$KEY" "$CMD" <<'PY'
import collections, csv, glob, json, sys
out, key, cmd = sys.argv[1:4]
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
res = {"command": cmd, "kernel_key": key, "launches": len(dur), "avg_ns": sum(dur) / max(1, len(dur))}
res.update({k: sum(v) / len(v) for k, v in agg.items()})
json.dump(res, open(out + "/pmc.json", "w"), indent=1)
print(json.dumps(res))
PY
