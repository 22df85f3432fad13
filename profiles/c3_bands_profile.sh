#!/bin/bash
# rocprofv3 evidence of C3's rank bands on ONE GPU (profiles/debug/c3_band_times.py: the band of every rank of a 1 / 2 / 4 / 8-rank split, one
# at a time): pass 1 --kernel-trace --stats, then one counter group per pass (bench.py's three groups).  Summary by band size (= by rank
# count): profiles/c3_bands_summary.py.  NOT a multi-GPU run -- what each rank's kernel would show, without any collective.
set -u
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/prof_r06_c3bands
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 profiles/debug/c3_band_times.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1 || echo "trace pass failed"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1 || echo "pmc pass $i ($grp) failed"
done
python3 profiles/c3_bands_summary.py "$OUT"
