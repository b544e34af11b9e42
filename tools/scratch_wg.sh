#!/bin/bash
run() { timeout -k 10 200 python bench.py --train-encoders --steps 10 --warmup 3 --repeats 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1', j['value'], j['ms_per_step'], j['timing']['ms_per_step_blocks'])
"; }
run b512 || exit 1
for d in UFND_WGRAD_BLOCKS=256 UFND_WGRAD_BLOCKS=384 UFND_WGRAD_BLOCKS=768; do
  python -m ultrafnd_git_amd.build --defs=$d --force > gpurun_out/wg_build.log 2>&1 || { echo "build failed $d"; continue; }
  run $d || echo "run failed"
done
python -m ultrafnd_git_amd.build --force > gpurun_out/wg_build.log 2>&1 || exit 1
run b512_b
