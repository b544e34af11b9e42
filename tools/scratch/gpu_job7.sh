set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k outlier > gpurun_out/r3/t8.log 2>&1; echo rc=$? >> gpurun_out/r3/t8.log; tail -3 gpurun_out/r3/t8.log | cut -c1-300
timeout -k 10 200 python tools/attn_bench.py 2>&1 | grep "B="
timeout -k 10 600 python -m ultrafnd_git_amd.build --defs=UFND_ATTN_KB128=1 > gpurun_out/r3/build_kb128.log 2>&1; echo build rc=$?
timeout -k 10 200 python tools/attn_bench.py 2>&1 | grep "B="
