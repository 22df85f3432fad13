#!/bin/bash
# The round's evidence set, taken ONCE on the final build (the set was re-taken six times in round 4).
#   bash profiles/run_round_evidence.sh <tag> a     tests, bench.py, rocprofv3 trace + PMC passes of the headline, the other configurations, C4 kernel trace
#   bash profiles/run_round_evidence.sh <tag> b     PMC passes of the grid form and of the sample-queue kernel, all-paths C2 parity, 150 000-scene soak
# Everything lands under gpurun_out/<tag>/ (and gpurun_out/prof_<tag>*/); profiles/summarize.py and the copies into profiles/ happen afterwards,
# in the build container.  Run through gpurun, one part per call.
set -u
TAG=${1:-r05}; PART=${2:-a}
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "$PART" = bench ] || [ "$PART" = headline ]; then   # (the bench step alone / with the headline's rocprofv3 passes)
  T0=$SECONDS; python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench.py wall seconds: $((SECONDS - T0))" | tee "$OUT/bench.time"
  if [ "$PART" = headline ]; then bash profiles/run_profile.sh "$TAG" > "$OUT/prof.log" 2>&1; fi
elif [ "$PART" = a ]; then
  python -m pytest tests -m gpu -q > "$OUT/tests.log" 2>&1; tail -n 3 "$OUT/tests.log"
  T0=$SECONDS; python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench.py wall seconds: $((SECONDS - T0))" | tee "$OUT/bench.time"
  bash profiles/run_profile.sh "$TAG" > "$OUT/prof.log" 2>&1
  python profiles/configs_bench.py --spp-c4 8 > "$OUT/configs.jsonl" 2> "$OUT/configs.err"; wc -l "$OUT/configs.jsonl"
  for r in "" "--retire"; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c4trace$r" -- python3 profiles/grid_bench.py --s 64 --reps 3 $r > "$OUT/c4trace$r.log" 2>&1 || echo "c4 trace $r failed"
  done
  find "$OUT" -name "*kernel_stats.csv" | head
else
  bash profiles/pmc_groups.sh "${TAG}_grid" queue8 "python3 profiles/grid_bench.py --s 16 --reps 2" \
    "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
    "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
    "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" > "$OUT/pmc_grid.log" 2>&1
  bash profiles/pmc_case.sh "${TAG}_queue" c2_retire > "$OUT/pmc_queue.log" 2>&1
  bash profiles/pmc_case.sh "${TAG}_queue_c5" c5_rr_retire > "$OUT/pmc_queue_c5.log" 2>&1
  python tests/full_size_c2_parity.py > "$OUT/c2_full_parity.log" 2>&1; tail -n 4 "$OUT/c2_full_parity.log"
  python profiles/debug/grid_form_soak.py --n 150000 --seed 20 > "$OUT/grid_form_soak.jsonl" 2> "$OUT/soak.err"; tail -n 2 "$OUT/grid_form_soak.jsonl"
fi
