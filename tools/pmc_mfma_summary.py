#!/usr/bin/env python3
"""MFMA utilisation per GEMM launch from one rocprofv3 PMC pass of bench.py:
   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
             --output-format csv -d <dir> -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --repeats 1
usage: pmc_mfma_summary.py <dir> <out.md>

SQ_VALU_MFMA_BUSY_CYCLES = matrix-pipe busy cycles summed over the chip's 1024 SIMDs (16 per v_mfma_f32_16x16x32_bf16: it equals
16 x the number of MFMA wave-instructions: column "MOPS" / 2 x 16).  MFMA utilisation = busy / (1024 SIMDs x launch duration x
2.4 GHz): the share of the matrix pipes' PEAK capacity used while the launch runs -- the same denominator as the 2.5 PFLOP/s
roofline.  (GRBM_GUI_ACTIVE / 8 / duration would give the clock actually held, but on dispatches this short it reads 2.7-4.8
"GHz": the counter includes ramp time around the dispatch -- MI355X_MICROARCH.md, DVFS give-back -- so it is listed, not used.)"""
import collections
import csv
import glob
import re
import sys

d, out = sys.argv[1], sys.argv[2]
f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
seen = set()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+(<[0-9, ]+>)?)", name)
    key = (m.group(1) if m else name[:50], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for k, v in agg.items():
    if "gemm_bf16" not in k[0] and "gemm_pp_kernel" not in k[0] and "attention" not in k[0]:
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    us = sum(dur[k]) / len(dur[k]) / 1e3
    rows.append((len(dur[k]) * us, k, len(dur[k]), us, cyc / us / 1e3, m["SQ_VALU_MFMA_BUSY_CYCLES"], m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * us * 2400.0),
                 m["SQ_INSTS_VALU_MFMA_MOPS_BF16"], m["SQ_BUSY_CYCLES"], m["SQ_WAVE_CYCLES"]))
rows.sort(reverse=True)
with open(out, "w") as fo:
    fo.write("# MFMA utilisation from PMC counters (rocprofv3, own pass)\n\n" + __doc__.split("usage")[0].strip().replace("\n", "  \n") + "\n\n")
    fo.write("| kernel | blocks | launches | avg us (this pass) | GRBM/8/us (not a clock) | MFMA busy cycles | **MFMA utilisation** | MFMA MOPS bf16 | SQ_BUSY_CYCLES | SQ_WAVE_CYCLES |\n|---|---|---|---|---|---|---|---|---|---|\n")
    tb = tc = 0.0
    for _, k, n, us, ghz, busy, util, mops, sqb, sqw in rows:
        fo.write(f"| {k[0]} | {k[1]} | {n} | {us:.2f} | {ghz:.2f} | {busy:.0f} | **{util:.3f}** | {mops:.0f} | {sqb:.0f} | {sqw:.0f} |\n")
        if "gemm_bf16" in k[0] or "gemm_pp_kernel" in k[0]:
            tb += busy * n
            tc += busy / util * n
    fo.write(f"\nAll `gemm_bf16_kernel` / `gemm_pp_kernel` launches, launch-weighted: MFMA utilisation **{tb / tc:.3f}** of the matrix pipes' cycles while a GEMM is running "
             "(counters serialise the streams: each launch has the chip to itself here).\n")
    fo.write("\nThe K loop itself issues MFMAs on 75-85 % of its cycles (in-kernel stamps: profiles/r03_gemm_stamps_g4.txt, earlier rounds' profiles/r0*_gemm_stamps*.txt); "
             "the launch-level figure is lower because prologue (first operands landing), epilogue (GELU, or residual + bf16 + statistics) and the tile counts "
             "that do not fill whole rounds of 256 CUs (the visual encoder's launches) are inside the launch duration.\n")
print(open(out).read())
