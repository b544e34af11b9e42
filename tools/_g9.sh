set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_gpu_trainer.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --seq-len 512 --batch 128 > gpurun_out/r2/h_L512.json 2> gpurun_out/r2/h_L512.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --frames 8 --batch 8 > gpurun_out/r2/h_F8.json 2> gpurun_out/r2/h_F8.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --batch 256 > gpurun_out/r2/h_B256.json 2> gpurun_out/r2/h_B256.err
python bench.py --head-only --steps 100 --warmup 10 --batch 256 > gpurun_out/r2/h_head_B256.json 2> gpurun_out/r2/h_head256.err
for f in h_L512 h_F8 h_B256 h_head_B256; do python - <<PY
import json
d=json.loads(open("gpurun_out/r2/$f.json").read().strip().splitlines()[-1]); r=d.get("roofline",{})
print("$f", d["value"], d["ms_per_step"], r.get("frac"), (r.get("per_launch") or {}).get("frac"))
PY
done
