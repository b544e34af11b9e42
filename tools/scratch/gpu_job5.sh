set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_tier_b_bwd.py tests/test_gpu_encoder_train.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/r3/t6.log 2>&1; echo rc=$? >> gpurun_out/r3/t6.log
tail -4 gpurun_out/r3/t6.log | cut -c1-300
timeout -k 10 300 python bench.py --train-encoders --steps 10 --warmup 3 --repeats 3 > gpurun_out/r3/te2.log 2> gpurun_out/r3/te2.err; echo rc=$? >> gpurun_out/r3/te2.log
python - <<'PY'
import json
for l in open("gpurun_out/r3/te2.log"):
    if l.startswith("{"):
        d=json.loads(l); print("train-encoders", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
