#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's headline kernel on the GPU box.
#   bash profiles/run_profile.sh <tag>        (run through gpurun; writes gpurun_out/prof_<tag>/)
# Pass 1: --kernel-trace --stats (per-kernel durations).  Passes 2..: --pmc only, one counter
# group per pass (never combined with tracing domains other than kernel-trace).
set -u
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
CMD="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-traffic-probe"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1 || echo "trace pass failed"
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1 || true
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i ($grp) failed"
done
find "$OUT" -name "*.csv" | head -50
