set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
true
run() { name=$1; shift; python bench.py --steps 40 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/r2/$name.json 2> gpurun_out/r2/$name.err || (tail -5 gpurun_out/r2/$name.err; exit 1); python - <<PY
import json
d=json.loads(open("gpurun_out/r2/$name.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$name", d["value"], d["ms_per_step"], r["frac"], {k:v["avg_us"] for k,v in r["by_shape_MxNxK"].items()})
PY
}
run b1_base
run b1_split --cu-split 192,64
run b1_split_vt --cu-split 192,64 --vis-tiles qkv=15,out=2,ffn1=22,ffn2=2
run b1_split_vt_tt --cu-split 192,64 --vis-tiles qkv=15,out=2,ffn1=22,ffn2=2 --text-tiles ffn1=15
run b1_split_vt2 --cu-split 192,64 --vis-tiles qkv=15,out=2,ffn1=15,ffn2=2
UFND_CU_MASK_LAYOUT=block run b1_split_vt_blk --cu-split 192,64 --vis-tiles qkv=15,out=2,ffn1=22,ffn2=2
run b1_base2
