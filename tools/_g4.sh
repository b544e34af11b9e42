set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_gpu_tier_b.py -m gpu -x -q -k "fused or attention or features" 2>&1 | tail -15
run() { name=$1; shift; python bench.py --steps 40 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/r2/$name.json 2> gpurun_out/r2/$name.err || (tail -5 gpurun_out/r2/$name.err; exit 1); python - <<PY
import json
d=json.loads(open("gpurun_out/r2/$name.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$name", d["value"], d["ms_per_step"], r["frac"], r["per_launch"]["avg_launch_us"], {k:v["avg_us"] for k,v in r["per_launch"]["by_shape_MxNxK"].items()})
PY
}
run c_fused



