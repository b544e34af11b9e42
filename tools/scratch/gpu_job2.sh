set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_tier_b_bwd.py tests/test_gpu_fullsize.py tests/test_gpu_tier_b.py -m gpu -q -x -s > gpurun_out/r3/t4.log 2>&1; echo rc=$? >> gpurun_out/r3/t4.log
grep -E "passed|failed|rc=|Error|error" gpurun_out/r3/t4.log | tail -15
