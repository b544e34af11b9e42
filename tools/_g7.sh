set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_gpu_tier_a.py tests/test_gpu_trainer.py tests/test_gpu_sizes.py -m gpu -x -q 2>&1 | tail -4
python bench.py --head-only --steps 200 --warmup 20 > gpurun_out/r2/f_head.json 2> gpurun_out/r2/f_head.err; cut -c1-330 gpurun_out/r2/f_head.json; echo
for i in 1 2 3; do
python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r2/f_full$i.json 2> gpurun_out/r2/f_full$i.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r2/f_full$i.json").read().strip().splitlines()[-1]); print("full$i", d["value"], d["ms_per_step"], d["timing"]["ms_per_step_blocks"])
PY
done
