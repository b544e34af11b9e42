set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_gpu_tier_b.py -m gpu -x -q -k "features or unpadded or fields" 2>&1 | tail -3
run() { name=$1; shift; python bench.py --steps 40 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/r2/$name.json 2> gpurun_out/r2/$name.err || (tail -5 gpurun_out/r2/$name.err; exit 1); python - <<PY
import json
d=json.loads(open("gpurun_out/r2/$name.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$name", d["value"], d["ms_per_step"], r["frac"], r["per_launch"]["avg_launch_us"], {k:v["avg_us"] for k,v in r["per_launch"]["by_shape_MxNxK"].items()})
PY
}
run d_base
run d_t_ffn1_15 --text-tiles ffn1=15
run d_v_ffn1_22 --vis-tiles ffn1=22
run d_v_qkv_17 --vis-tiles qkv=17
run d_v_all256 --vis-tiles qkv=22,ffn1=22,out=2,ffn2=2
run d_t_out2 --text-tiles out=2,ffn2=2
run d_base2
