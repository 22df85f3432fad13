#!/bin/bash
# Per-rank rocprofv3 evidence of `bench.py --gpus N` (N > 1) on ONE node: every rank is its own `rocprofv3 ... -- python3 bench.py ...` process
# (the profiler wraps the program directly; torchrun starts the ranks as children before anything touches a GPU).
#   bash profiles/run_profile_multi.sh N [out-dir] [extra bench.py flags]
# Passes, one torchrun launch each: (1) --kernel-trace --stats -> <out>/stats/rank<r>/; (2..) --kernel-trace --pmc <group> (no other tracing
# domain), one counter group per pass -> <out>/pmc<i>/rank<r>/.  The default groups are bench.py's own (PMC_PASSES: FETCH_SIZE | WRITE_SIZE |
# the VALU group), so that the summary -- profiles/multi_summary.py, <out>/summary.json, one line per rank -- reports per GPU what north_star
# asks for: the band kernel's duration, achieved HBM GB/s, VALU instructions per SIMD-cycle, resident waves per SIMD and lane activity.
# APT_PROF_PMC="<group>;<group>..." replaces the groups; APT_PROF_PMC=none runs the stats pass only.
# On a one-GPU box this can only be rehearsed with APT_BENCH_SHARE_GPU=1 (ranks share the card, gloo instead of RCCL: the durations then
# measure nothing).  North star: "rocprof ... at 1/2/4/8 GPUs" -- N = 1 is profiles/run_profile.sh.
set -u
N=${1:?number of GPUs}
OUT=${2:-gpurun_out/prof_multi_$N}
shift; shift || true
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p "$OUT"
cat > "$OUT/rank.sh" <<'EOS'
#!/bin/bash
if [ -n "${APT_PROF_GROUP:-}" ]; then
  exec rocprofv3 --kernel-trace --pmc $APT_PROF_GROUP --output-format csv -d "$APT_PROF_OUT/rank$RANK" -- python3 bench.py "$@"
fi
exec rocprofv3 --kernel-trace --stats --output-format csv -d "$APT_PROF_OUT/rank$RANK" -- python3 bench.py "$@"
EOS
chmod +x "$OUT/rank.sh"
launch() {   # launch <sub-directory> <log>   (APT_PROF_GROUP in the environment selects the pass)
  APT_PROF_OUT="$OUT/$1" timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 \
      --master-port 29541 --no-python "$OUT/rank.sh" --gpus "$N" --steps 5 --warmup 2 "${EXTRA[@]}" > "$OUT/$2" 2>&1
}
EXTRA=("$@")
unset APT_PROF_GROUP
launch stats bench.log
rc=$?
grep '^{' "$OUT/bench.log" > "$OUT/bench.json" || true
GROUPS_STR=${APT_PROF_PMC:-$(python3 -c "import bench; print(';'.join(' '.join(g) for g in bench.PMC_PASSES))")}
if [ "$rc" -eq 0 ] && [ "$GROUPS_STR" != "none" ]; then
  i=0
  IFS=';' read -ra GRPS <<< "$GROUPS_STR"
  for grp in "${GRPS[@]}"; do
    i=$((i+1))
    echo "pmc pass $i: $grp"
    APT_PROF_GROUP="$grp" launch "pmc$i" "pmc$i.log" || echo "pmc pass $i ($grp) failed"
  done
fi
python3 profiles/multi_summary.py "$OUT" "$N"
exit $rc
