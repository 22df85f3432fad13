#!/usr/bin/env python3
"""Condense what `run_round_evidence.sh <tag> a` and `... b` left under gpurun_out/ into the committed files profiles/<tag>_*:

    python profiles/collect_round.py r06

a: summarize.py (kernel stats, PMC summary incl. the instruction-cache figure, hbm_traffic.json, the bench line), the configurations,
   the C4 kernel traces, the kernel resource table of the build in the tree;
b: the grid form's PMC passes with their derived figures, the sample-queue kernel's PMC passes, the all-paths C2 parity log, the soak.
Parts that are missing are skipped and named.  Run in the build container after gpurun has merged the outputs back."""
import glob
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(here)
out = os.path.join(root, "gpurun_out", tag)
done, missing = [], []


def have(path):
    return os.path.exists(path) and os.path.getsize(path) > 0


# ---- part a ------------------------------------------------------------------------------------------------------------------------
bench = os.path.join(out, "bench.json")
if have(bench) and glob.glob(os.path.join(root, "gpurun_out", f"prof_{tag}", "trace", "*", "*_kernel_stats.csv")):
    subprocess.run([sys.executable, os.path.join(here, "summarize.py"), tag, bench], check=True, cwd=root, stdout=subprocess.DEVNULL)
    done.append("summarize.py (kernel stats, pmc, hbm_traffic, bench line)")
else:
    missing.append("bench.json / prof trace")
if have(os.path.join(out, "configs.jsonl")):
    shutil.copy(os.path.join(out, "configs.jsonl"), os.path.join(here, f"{tag}_configs.jsonl"))
    done.append("configs")
else:
    missing.append("configs.jsonl")
traces = [(r, glob.glob(os.path.join(out, f"c4trace{r}", "*", "*_kernel_stats.csv"))) for r in ("", "--retire")]
if all(t for _, t in traces):
    with open(os.path.join(here, f"{tag}_c4_kernel_stats.csv"), "w") as f:
        for r, t in traces:
            f.write(f"# rocprofv3 --kernel-trace --stats -- python3 profiles/grid_bench.py --s 64 --reps 3 {r}\n".replace(" \n", "\n"))
            f.write(open(max(t, key=os.path.getmtime)).read())
    done.append("C4 kernel traces")
else:
    missing.append("c4trace")
r = subprocess.run([sys.executable, os.path.join(here, "kernel_resources.py")], cwd=root, capture_output=True, text=True)
if r.returncode == 0 and r.stdout.strip():
    open(os.path.join(here, f"{tag}_kernel_resources.txt"), "w").write(r.stdout)
    done.append("kernel resources (needs csrc/render_kernels.s: make -C ascendpathtracing_amd/csrc asm)")

# ---- part b ------------------------------------------------------------------------------------------------------------------------
gp = os.path.join(root, "gpurun_out", f"prof_{tag}_grid", "pmc.json")
if have(gp):
    p = json.load(open(gp))
    seg = 1920 * 1080 * 4 * 16 * 8                                     # grid_bench.py --s 16: 64 spp, depth 8
    cyc = p["GRBM_GUI_ACTIVE"] / 8
    d = {"kernel_ms": p["avg_ns"] / 1e6,
         "valu_lane_activity": p["SQ_THREAD_CYCLES_VALU"] / (p["SQ_ACTIVE_INST_VALU"] * 64),
         "valu_insts_x64_per_lane_segment": p["SQ_INSTS_VALU"] * 64 / seg,
         "valu_insts_per_simd_cycle": p["SQ_INSTS_VALU"] / (cyc * 1024),
         "salu_insts_per_valu_inst": p["SQ_INSTS_SALU"] / p["SQ_INSTS_VALU"],
         "valu_share_of_issued_instructions": p["SQ_INSTS_VALU"] / p["SQ_ACTIVE_INST_ANY"],
         "ta_busy_fraction": p["TA_TA_BUSY_sum"] / (cyc * 256),
         "ta_cycles_per_wave_level_load": p["TA_TA_BUSY_sum"] / p["TA_TOTAL_WAVEFRONTS_sum"],
         "wave_level_loads_per_64_lane_segments": p["TA_TOTAL_WAVEFRONTS_sum"] / (seg / 64),
         "mean_waves_per_simd": p["SQ_WAVE_CYCLES"] * 4 / (cyc * 1024),
         "fraction_of_wave_time_waiting_on_a_counter": p["SQ_WAIT_ANY"] / p["SQ_WAVE_CYCLES"]}
    json.dump({"what": "C4 scene (10 000 spheres, 1080p, 64 spp, depth 8) through the sample-queue kernel's grid form (pt_queue.h run_grid): "
                       "three rocprofv3 counter passes (profiles/pmc_groups.sh) and what follows from them",
               "command": p.get("command"), "pmc": p, "derived": d}, open(os.path.join(here, f"{tag}_grid_queue_pmc.json"), "w"), indent=1)
    done.append("grid form PMC")
else:
    missing.append("prof_%s_grid/pmc.json" % tag)
for case, name in (("queue", "queue_pmc"), ("queue_c5", "queue_c5_pmc")):
    src = os.path.join(root, "gpurun_out", f"prof_{tag}_{case}", "pmc.json")
    if have(src):
        shutil.copy(src, os.path.join(here, f"{tag}_{name}.json"))
        done.append(name)
    else:
        missing.append(src)
if have(os.path.join(out, "c2_full_parity.log")):
    shutil.copy(os.path.join(out, "c2_full_parity.log"), os.path.join(here, f"{tag}_c2_full_parity.log"))
    done.append("all-paths C2 parity log")
else:
    missing.append("c2_full_parity.log")
soak = os.path.join(out, "grid_form_soak.jsonl")
if have(soak):
    open(os.path.join(here, f"{tag}_grid_form_soak.json"), "w").write(open(soak).read().strip().splitlines()[-1] + "\n")
    done.append("soak")
else:
    missing.append("grid_form_soak.jsonl")
print("collected:", "; ".join(done))
if missing:
    print("MISSING:", "; ".join(missing))
