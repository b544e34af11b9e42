#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE in their own
runs, as the MI355X guide prescribes): reads = 2 x FETCH_SIZE KiB (gfx950 counts 128-B requests at 64 B),
writes = WRITE_SIZE KiB.   usage: pmc_summary.py <fetch_dir> <write_dir> <out.md>"""
import collections
import csv
import glob
import re
import sys


def load(d, counter):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        m = re.match(r"([A-Za-z0-9_]+(<[0-9, ]+>)?)", name)
        key = (m.group(1) if m else name[:50], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        agg[key].append(float(r["Counter_Value"]))
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for k in fetch:
    rd = 2 * sum(fetch[k]) / len(fetch[k]) * 1024 / 1e6
    wr = sum(write.get(k, [0])) / max(1, len(write.get(k, [0]))) * 1024 / 1e6
    rows.append((rd * len(fetch[k]), k, len(fetch[k]), rd, wr))
rows.sort(reverse=True)
with open(sys.argv[3], "w") as f:
    f.write("# HBM traffic from PMC counters (rocprofv3, separate passes)\n\n"
            "Commands (GPU box): `rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ... -- python3 bench.py --steps 4 "
            "--warmup 2 --no-cpu-baseline` and the same with `--pmc WRITE_SIZE`.  Counters are KiB; on gfx950 FETCH_SIZE reports half of a "
            "wide coalesced read, so reads = 2 x FETCH_SIZE x 1024 B; WRITE_SIZE is exact (the AdamW row calibrates both: 4 x 51.0 MB read, "
            "3 x 51.0 MB written).\n\n| kernel | blocks | launches | HBM read MB/launch (2xFETCH) | HBM written MB/launch |\n|---|---|---|---|---|\n")
    for _, k, n, rd, wr in rows[:40]:
        f.write(f"| {k[0]} | {k[1]} | {n} | {rd:.1f} | {wr:.1f} |\n")
print(open(sys.argv[3]).read()[:3000])
