set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2/xcd2d
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_tier_b.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/b$i.json 2> $O/b$i.err; python - <<PY
import json
d=json.loads(open("$O/b$i.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("2d run$i", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in r["per_launch"]["by_shape_MxNxK"].items()})
PY
done
rm -rf $O/pmc_fetch
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --repeats 1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
python3 - <<PY
import csv, glob, collections, re
f = glob.glob("$O/pmc_fetch/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE" or "gemm_bf16" not in r["Kernel_Name"]: continue
    m = re.search(r"gemm_bf16_kernel<([0-9, ]+)>", r["Kernel_Name"])
    agg[(m.group(1), int(r["Grid_Size"]) // int(r["Workgroup_Size"]))].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items(), key=lambda kv: -len(kv[1])):
    print(k, len(v), round(2 * sum(v) / len(v) * 1024 / 1e6, 1), "MB read/launch")
PY
