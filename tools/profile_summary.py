#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run of bench.py into profiles/<tag>_summary.md.
usage: profile_summary.py <trace_dir> <bench_log> <tag> [passes]"""
import collections
import csv
import glob
import json
import shutil
import sys

trace_dir, bench_log, tag = sys.argv[1], sys.argv[2], sys.argv[3]
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 28
tf = glob.glob(f"{trace_dir}/*/*_kernel_trace.csv")[0]
sf = glob.glob(f"{trace_dir}/*/*_kernel_stats.csv")[0]
shutil.copy(sf, f"profiles/{tag}_kernel_stats.csv")
rows = list(csv.DictReader(open(tf)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
j = json.loads([l for l in open(bench_log) if l.startswith('{"metric"')][-1])


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:56]


agg = collections.defaultdict(list)
for r in rows:
    key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
gem = [r for r in rows if "gemm_bf16" in r["Kernel_Name"]]
n_inst = 3 * j["roofline"]["launches_per_step"]
inst = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in gem[-n_inst:]]
over = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in gem[:-n_inst]]
with open(f"profiles/{tag}_summary.md", "w") as f:
    f.write(f"# {tag}\n\nCommand (GPU box, 1x MI355X): `cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv "
            f"-d gpurun_out/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline`\n\n")
    f.write(f"bench line under the profiler: **{j['value']} samples/s, {j['ms_per_step']} ms/step**; roofline object: `{json.dumps(j['roofline'])}`\n\n")
    f.write(f"`gemm_bf16_kernel` launch durations (rocprofv3): instrumented sequential pass (the one bench.py times with HIP events): "
            f"{len(inst)} launches, average **{sum(inst) / len(inst) / 1e3:.2f} us** (HIP events in the same run: {j['roofline']['avg_launch_us']} us); "
            f"timed region (three concurrent streams, kernels share the CUs): {len(over)} launches, average {sum(over) / max(1, len(over)) / 1e3:.2f} us.\n\n")
    f.write("| kernel | blocks (x,y,z) | launches | median us | min us | total ms |\n|---|---|---|---|---|---|\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:45]:
        v2 = sorted(v)
        f.write(f"| {k[0]} | {k[1]},{k[2]},{k[3]} | {len(v)} | {v2[len(v2) // 2] / 1e3:.2f} | {v2[0] / 1e3:.2f} | {sum(v) / 1e6:.3f} |\n")
    tot = sum(sum(v) for v in agg.values())
    f.write(f"\nAll kernels: {tot / 1e6:.2f} ms over {passes} encoder passes.\n")
print(open(f"profiles/{tag}_summary.md").read()[:1500])
