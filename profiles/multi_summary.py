#!/usr/bin/env python3
"""Per-rank summary of a `profiles/run_profile_multi.sh` directory: what north_star asks the rocprof evidence of an N-GPU run to show --
achieved HBM GB/s and VALU occupancy per GPU -- next to the band kernel's duration.

    python3 profiles/multi_summary.py <out-dir> <N>      -> one line per rank on stdout, <out-dir>/summary.json

Layout it reads (written by run_profile_multi.sh): <out>/stats/rank<r>/ (rocprofv3 --kernel-trace --stats) and <out>/pmc<i>/rank<r>/ (one
counter group per pass; FETCH_SIZE and WRITE_SIZE in passes of their own, as MI355X_MICROARCH.md prescribes), <out>/bench.json (rank 0's
line: segments per GPU).  The formulas are bench.py's own (derive_valu / the traffic conversion: KiB, FETCH_SIZE x2 on gfx950), the launches
are matched by dispatch id and every process's FIRST launch is dropped, so a rank's figures here and `roofline` in a one-GPU bench line mean
the same thing.  No GPU needed: tests/test_bench_dry_run.py feeds it stand-in CSVs."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

KERNEL = "render_frame_kernel<"       # the band kernel of whatever workload the ranks ran (C3: the same instantiation as the headline)


def rank_summary(out, r, segments):
    avg, ns = {}, {}
    for d in sorted(glob.glob(os.path.join(out, "pmc*", f"rank{r}"))):
        vals, dur = bench.counter_rows(d, kernel=KERNEL)
        for c, v in vals.items():
            if v:
                avg[c] = sum(v) / len(v)
                ns[c] = sum(dur) / max(1, len(dur))
    res = {"rank": r}
    for f in glob.glob(os.path.join(out, "stats", f"rank{r}", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if KERNEL in row["Name"]:
                res.update({"kernel": row["Name"][:90], "calls": int(row["Calls"]), "avg_ms": round(float(row["AverageNs"]) / 1e6, 3),
                            "pct_of_gpu_time": float(row["Percentage"])})
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        traffic = 2.0 * avg["FETCH_SIZE"] * 1024.0 + avg["WRITE_SIZE"] * 1024.0
        res.update({"traffic_bytes_per_launch": round(traffic), "hbm_gbps": round(traffic / ns["WRITE_SIZE"], 3)})   # bytes / ns = GB/s
    if all(c in avg for c in ("GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES")):
        res.update(bench.derive_valu(avg, ns["SQ_INSTS_VALU"], segments or 1))
        if not segments:
            res.pop("valu_insts_per_segment", None)
    return res


def main(out, n):
    segments = None
    try:
        with open(os.path.join(out, "bench.json")) as f:
            line = [l for l in f if l.startswith("{")]
        segments = json.loads(line[-1])["config"]["segments_per_gpu"] if line else None
    except (OSError, ValueError, KeyError):
        pass
    ranks = [rank_summary(out, r, segments) for r in range(n)]
    for s in ranks:
        print("rank %d: " % s["rank"] + "  ".join(f"{k} {v}" for k, v in s.items() if k not in ("rank", "kernel")))
    with open(os.path.join(out, "summary.json"), "w") as f:
        json.dump({"n_gpus": n, "segments_per_gpu": segments, "ranks": ranks}, f, indent=1)
    return ranks


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
