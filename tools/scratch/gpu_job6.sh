set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_tier_b.py tests/test_gpu_tier_b_bwd.py tests/test_gpu_tier_a.py tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/r3/t7.log 2>&1; echo rc=$? >> gpurun_out/r3/t7.log
tail -4 gpurun_out/r3/t7.log | cut -c1-300
timeout -k 10 200 python tools/attn_bench.py > gpurun_out/r3/attn_bench.log 2>&1; cat gpurun_out/r3/attn_bench.log | grep "B="
timeout -k 10 300 python bench.py --seq-len 512 --batch 128 --steps 8 --warmup 4 --repeats 3 --no-cpu-baseline > gpurun_out/r3/L512_b.json 2> gpurun_out/r3/L512_b.err
timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/r3/main_b.json 2> gpurun_out/r3/main_b.err
python - <<'PY'
import json
for f in ("L512_b", "main_b"):
    for l in open(f"gpurun_out/r3/{f}.json"):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("lookahead_1", {}) and d["lookahead_1"]["value"])
PY
