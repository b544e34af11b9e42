set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_encoder_train.py tests/test_gpu_fullsize.py tests/test_gpu_tier_b.py -m gpu -q -x -s > gpurun_out/r3/t5.log 2>&1; echo rc=$? >> gpurun_out/r3/t5.log
grep -E "passed|failed|rc=|Error|error|worst relative|grad norm|outlier-shaped" gpurun_out/r3/t5.log | cut -c1-300 | tail -25
