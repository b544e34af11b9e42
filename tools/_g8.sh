set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
run() { name=$1; shift; python bench.py --steps 40 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/r2/$name.json 2> gpurun_out/r2/$name.err || (tail -5 gpurun_out/r2/$name.err; exit 1); python - <<PY
import json
d=json.loads(open("gpurun_out/r2/$name.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$name", d["value"], d["ms_per_step"], {k:v["avg_us"] for k,v in r["per_launch"]["by_shape_MxNxK"].items()})
PY
}
run g_base
run g_fit1 --text-tiles ffn1=15 --vis-tiles qkv=15,out=2,ffn2=2,ffn1=15
run g_fit2 --text-tiles ffn1=15 --vis-tiles qkv=15,out=2,ffn2=2
run g_fit3 --text-tiles ffn1=15 --vis-tiles qkv=15
run g_fit4 --vis-tiles qkv=15,out=2,ffn2=2,ffn1=15
run g_base2
