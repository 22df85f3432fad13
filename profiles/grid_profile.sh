#!/bin/bash
# PMC passes for the grid traversal kernel (profiles/grid_bench.py).  bash profiles/grid_profile.sh <tag>
set -u
TAG=${1:-grid}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
CMD="python3 profiles/grid_bench.py --s 16 --reps 2"
python3 profiles/grid_bench.py --s 16 --reps 3 --stats > "$OUT/bench.json" 2>"$OUT/bench.err" || echo "bench failed"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i ($grp) failed"
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/pmc*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "render_frame_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in agg.items()}
json.dump(res, open(out + "/pmc.json", "w"), indent=1)
print(json.dumps(res))
PY
cat "$OUT/bench.json"
