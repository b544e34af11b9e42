#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run of `bench.py --head-only` (the reference's own training mode: cached
features, no encoders in the step).   usage: head_profile_summary.py <trace_dir> <profiled.json> <unprofiled.json> <tag> <steps_in_trace>"""
import collections
import csv
import glob
import json
import shutil
import sys

trace_dir, prof_json, unprof_json, tag, steps = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
tf = glob.glob(f"{trace_dir}/*/*_kernel_trace.csv")[0]
shutil.copy(glob.glob(f"{trace_dir}/*/*_kernel_stats.csv")[0], f"profiles/{tag}_kernel_stats.csv")
rows = sorted(csv.DictReader(open(tf)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]


# steady state only: the launches between the (steps - 40)-th and the last optimizer kernel (warm-up, graph capture and
# the buffers' initial copies are before that)
opt = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
keep = min(40, len(opt) - 1)
rows = rows[opt[-keep - 1] + 1:opt[-1] + 1]
steps = keep
agg = collections.defaultdict(list)
for r in rows:
    agg[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * int(r["Grid_Size_Y"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
jp = json.loads([l for l in open(prof_json) if l.startswith('{"metric"')][-1])
ju = json.loads([l for l in open(unprof_json) if l.startswith('{"metric"')][-1])
shutil.copy(unprof_json, f"profiles/{tag}_unprofiled.json")
with open(f"profiles/{tag}_summary.md", "w") as f:
    B = ju.get("per_gpu_batch", 32)
    f.write(f"# {tag}: head-only step (cached features), B = {B}\n\nCommand: `rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python3 bench.py "
            f"--head-only{'' if B == 32 else f' --batch {B}'} --steps {jp['steps']} --warmup {jp['warmup']} --repeats 1`; the table covers the last {steps} steps of the trace (steady state).\n\n")
    f.write(f"Unprofiled, same box: **{ju['value']} samples/s, {ju['ms_per_step']} ms/step**, {ju['roofline']['algorithmic_MB_per_step']} MB algorithmic "
            f"=> **{ju['roofline']['achieved']} GB/s = {ju['roofline']['frac']} of 8 TB/s** (`profiles/{tag}_unprofiled.json`).  Under the profiler: "
            f"{jp['ms_per_step']} ms/step.\n\n")
    f.write("| kernel | blocks | launches per step | median us | total us per step |\n|---|---|---|---|---|\n")
    tot = 0.0
    nl = 0.0
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        per = len(v) / steps
        if per < 0.5:
            continue
        v2 = sorted(v)
        f.write(f"| {k[0]} | {k[1]} | {per:.2f} | {v2[len(v2) // 2] / 1e3:.2f} | {sum(v) / steps / 1e3:.1f} |\n")
        tot += sum(v) / steps / 1e3
        nl += per
    f.write(f"\n{nl:.1f} launches per step, {tot:.0f} us of kernel time per step (profiled clock); the step is a serial chain, so kernel time + "
            f"{nl:.0f} launch boundaries is its duration.\n")
print(open(f"profiles/{tag}_summary.md").read())
