#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run of bench.py into profiles/<tag>_summary.md.
usage: profile_summary.py <trace_dir> <bench_log> <tag> [passes] [unprofiled.json] [bench arguments of the traced command]"""
import collections
import csv
import glob
import json
import shutil
import sys

trace_dir, bench_log, tag = sys.argv[1], sys.argv[2], sys.argv[3]
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 28
unprofiled = sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] != "-" else None
bench_args = sys.argv[6] if len(sys.argv) > 6 else "--steps 20 --warmup 5 --no-cpu-baseline --no-lookahead-compare --repeats 1"
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
gem = [r for r in rows if "gemm_bf16" in r["Kernel_Name"] or "gemm_pp_kernel" in r["Kernel_Name"]]
pl = j["roofline"].get("per_launch")          # (absent in the --train-encoders line: its roofline is the whole step only)
n_inst = 3 * j["roofline"].get("launches_per_encoder_pass", j["roofline"].get("launches_per_step", 0)) if pl else 0
inst = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in gem[-n_inst:]] if n_inst else []
over = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in (gem[:-n_inst] if n_inst else gem)]
with open(f"profiles/{tag}_summary.md", "w") as f:
    f.write(f"# {tag}\n\nCommand (GPU box, 1x MI355X): `cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv "
            f"-d gpurun_out/prof -- python3 bench.py {bench_args}` "
            f"(default encoder lookahead: {j['config'].get('encoder_lookahead_batches', 1)} batches per encoder pass)\n\n")
    f.write(f"bench line UNDER THE PROFILER (the profiler slows the step; never compare it with an unprofiled run): "
            f"**{j['value']} samples/s, {j['ms_per_step']} ms/step**, whole-step `roofline.frac` {j['roofline']['frac']}.\n\n")
    if pl:
        roc = sum(inst) / len(inst) / 1e3
        f.write(f"Cross-check of the per-launch figure, both taken in THIS run: `gemm_bf16_kernel` + `gemm_pp_kernel` durations in the instrumented sequential pass "
                f"(the one bench.py times with HIP events): rocprofv3 {len(inst)} launches, average **{roc:.2f} us**; HIP events minus marker price in the "
                f"same pass: **{pl['avg_launch_us']} us** (raw event interval {pl['avg_event_interval_us']} us, marker {pl['event_marker_us']} us) -- "
                f"{abs(pl['avg_launch_us'] / roc - 1) * 100:.1f} % apart.  In the timed region (three concurrent streams, kernels share the CUs) rocprofv3 "
                f"sees {len(over)} launches, average {sum(over) / max(1, len(over)) / 1e3:.2f} us.\n\n")
        fl = j["roofline"]["flops_per_launch_avg"]
        f.write(f"Reproducing `roofline` from this file: per launch {fl / 1e9:.3f} GFLOP / {roc:.2f} us = {fl / roc / 1e6:.0f} TFLOP/s = {fl / roc / 1e6 / 2500:.3f} of "
                f"2.5 PFLOP/s (profiled clock); whole step {j['roofline']['flops_per_step'] / 1e9:.1f} GFLOP / {j['ms_per_step']} ms = "
                f"{j['roofline']['flops_per_step'] / j['ms_per_step'] / 1e9:.0f} TFLOP/s = {j['roofline']['frac']}.\n\n")
    else:
        f.write(f"`gemm_bf16_kernel` + `gemm_pp_kernel` launches in the trace (forward and backward forms): {len(over)}, average {sum(over) / max(1, len(over)) / 1e3:.2f} us.\n\n")
    if unprofiled:
        u = json.loads([l for l in open(unprofiled) if l.startswith('{"metric"')][-1])
        up = u["roofline"].get("per_launch")
        f.write(f"The same command WITHOUT the profiler, same box, minutes earlier (`profiles/{tag}_unprofiled.json`): **{u['value']} samples/s, "
                f"{u['ms_per_step']} ms/step** (blocks {u.get('timing', {}).get('ms_per_step_blocks')}), whole-step frac **{u['roofline']['frac']}**"
                + (f", per-launch {up['avg_launch_us']} us = {up['achieved']} TFLOP/s ({up['frac']})" if up else "")
                + (f"; the same optimizer steps with one batch per encoder pass, same process: {u['lookahead_1']['value']} samples/s, {u['lookahead_1']['ms_per_step']} ms/step"
                   if u.get("lookahead_1") else "") + ".\n\n")
        shutil.copy(unprofiled, f"profiles/{tag}_unprofiled.json")
    if pl:
        f.write("Per shape (HIP events, this profiled run): " + json.dumps(pl["by_shape_MxNxK"]) + "\n\n")
    f.write("| kernel | blocks (x,y,z) | launches | median us | min us | total ms |\n|---|---|---|---|---|---|\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:45]:
        v2 = sorted(v)
        f.write(f"| {k[0]} | {k[1]},{k[2]},{k[3]} | {len(v)} | {v2[len(v2) // 2] / 1e3:.2f} | {v2[0] / 1e3:.2f} | {sum(v) / 1e6:.3f} |\n")
    tot = sum(sum(v) for v in agg.values())
    f.write(f"\nAll kernels: {tot / 1e6:.2f} ms over {passes} encoder passes.\n")
print(open(f"profiles/{tag}_summary.md").read()[:1500])
