#!/bin/bash
# Per-rank rocprofv3 kernel statistics of `bench.py --gpus N` (N > 1) on ONE node: every rank is its own
# `rocprofv3 --kernel-trace --stats -- python3 bench.py ...` process (the profiler wraps the program directly; torchrun
# starts the ranks as children before anything touches a GPU).  Writes <out>/rank<r>/ and prints one line per rank:
# the band kernel's calls, average and total duration.
#   bash profiles/run_profile_multi.sh N [out-dir] [extra bench.py flags]
# With APT_PROF_PMC="<counters>" in the environment the ranks collect those counters instead (`--kernel-trace --pmc ...`, no
# other tracing domain; one counter group per call, e.g. "FETCH_SIZE", then "WRITE_SIZE", then
# "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE") and the script prints per rank the per-launch averages for the band kernel.
# On a one-GPU box this can only be rehearsed with APT_BENCH_SHARE_GPU=1 (ranks share the card, gloo instead of RCCL:
# the durations then measure nothing).  North star: "rocprof ... at 1/2/4/8 GPUs" -- N = 1 is profiles/run_profile.sh.
set -u
N=${1:?number of GPUs}
OUT=${2:-gpurun_out/prof_multi_$N}
shift; shift || true
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p "$OUT"
cat > "$OUT/rank.sh" <<'EOS'
#!/bin/bash
if [ -n "${APT_PROF_PMC:-}" ]; then
  exec rocprofv3 --kernel-trace --pmc $APT_PROF_PMC --output-format csv -d "$APT_PROF_OUT/rank$RANK" -- python3 bench.py "$@"
fi
exec rocprofv3 --kernel-trace --stats --output-format csv -d "$APT_PROF_OUT/rank$RANK" -- python3 bench.py "$@"
EOS
chmod +x "$OUT/rank.sh"
APT_PROF_OUT="$OUT" timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 \
    --master-port 29541 --no-python "$OUT/rank.sh" --gpus "$N" --steps 5 --warmup 2 "$@" > "$OUT/bench.log" 2>&1
rc=$?
grep '^{' "$OUT/bench.log" > "$OUT/bench.json" || true
python3 - "$OUT" "$N" <<'EOP'
import csv, glob, sys
out, n = sys.argv[1], int(sys.argv[2])
import collections
for r in range(n):
    for f in glob.glob(f"{out}/rank{r}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "render_frame_kernel" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(f"rank {r}: " + "  ".join(f"{k} {sum(v) / len(v):.6g} per launch ({len(v)} launches)" for k, v in sorted(agg.items())))
    for f in glob.glob(f"{out}/rank{r}/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "render_frame_kernel" in row["Name"]:
                print(f"rank {r}: {row['Name'][:80]}  calls {row['Calls']}  "
                      f"avg {float(row['AverageNs']) / 1e6:.3f} ms  total {float(row['TotalDurationNs']) / 1e6:.1f} ms  {row['Percentage']} % of GPU time")
EOP
exit $rc
