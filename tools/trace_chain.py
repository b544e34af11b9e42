#!/usr/bin/env python3
"""Per-kernel medians of the second half of a rocprofv3 --kernel-trace run, in first-appearance order (the launch chain of
one steady-state step), with launches per step.   usage: trace_chain.py <trace_dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(f"{sys.argv[1]}/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48],
         int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))
    agg.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = []
for k, v in agg.items():
    v.sort()
    out.append((len(v), k, v[len(v) // 2] / 1e3))
n = min(c for c, _, _ in out if c > 5)
for c, k, m in out:
    print(f"{c:5d} x{c / n:4.1f} {m:7.2f} us  {k[0]} [{k[1]}]")
print(f"kernel time per step: {sum(c / n * m for c, k, m in out if c >= n):.1f} us; span per step: "
      f"{(int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3 / (sum(c for c, _, _ in out) / sum(c / n for c, _, _ in out if c >= n)):.1f} us")
