set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
python tools/cu_mask_probe.py > gpurun_out/r2/probe.json 2> gpurun_out/r2/probe.err || (tail -20 gpurun_out/r2/probe.err; exit 1)
cat gpurun_out/r2/probe.json
python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r2/b0_base.json 2> gpurun_out/r2/b0.err
python - <<'PY'
import json
for f in ["b0_base"]:
    d=json.loads(open(f"gpurun_out/r2/{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
