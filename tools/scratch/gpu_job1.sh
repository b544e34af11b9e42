set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 300 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_dp.py tests/test_bench_launch.py -m gpu -q -x > gpurun_out/r3/t3.log 2>&1; echo rc=$? >> gpurun_out/r3/t3.log
timeout -k 10 600 python tools/scratch/measure_parity.py > gpurun_out/r3/parity.log 2>&1; echo rc=$? >> gpurun_out/r3/parity.log
for rd in fp32 bf16 fp32 bf16; do
  timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-lookahead-compare --residual-dtype $rd >> gpurun_out/r3/ab_resid.log 2>gpurun_out/r3/ab_resid.err || echo "fail $rd" >> gpurun_out/r3/ab_resid.log
done
tail -3 gpurun_out/r3/t3.log; tail -12 gpurun_out/r3/parity.log; python - <<'PY'
import json
for l in open("gpurun_out/r3/ab_resid.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["residual_stream"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["per_launch"]["by_shape_MxNxK"].get("16384x768x768"))
PY
