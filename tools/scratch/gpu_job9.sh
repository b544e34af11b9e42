set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_sizes.py tests/test_gpu_tier_a.py tests/test_gpu_trainer.py tests/test_gpu_integrated.py -m gpu -q -x > gpurun_out/r3/t10.log 2>&1; echo rc=$? >> gpurun_out/r3/t10.log
tail -4 gpurun_out/r3/t10.log | cut -c1-300
for b in 32 256 128; do timeout -k 10 200 python bench.py --head-only --batch $b --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('head-only B', d['per_gpu_batch'], d['value'], d['ms_per_step'])
"; done
