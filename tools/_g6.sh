set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python bench.py --head-only --steps 200 --warmup 20 > gpurun_out/r2/e_head.json 2> gpurun_out/r2/e_head.err; cat gpurun_out/r2/e_head.json | cut -c1-400
python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r2/e_full.json 2> gpurun_out/r2/e_full.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r2/e_full.json").read().strip().splitlines()[-1]); print("full", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
export TMPDIR=/tmp
rm -rf gpurun_out/r2/prof_head2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_head2 -- python3 bench.py --head-only --steps 50 --warmup 10 --repeats 1 > gpurun_out/r2/e_head_profiled.json 2> gpurun_out/r2/e_head_profiled.err
