set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 300 python bench.py --train-encoders --steps 10 --warmup 3 --repeats 3 > gpurun_out/r3/te.log 2> gpurun_out/r3/te.err; echo rc=$? >> gpurun_out/r3/te.log
tail -3 gpurun_out/r3/te.log | cut -c1-900; tail -5 gpurun_out/r3/te.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3/prof_te -- python3 $GRAFT_REPO_ROOT/bench.py --train-encoders --steps 6 --warmup 2 --repeats 1 > $GRAFT_REPO_ROOT/gpurun_out/r3/te_prof.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT && f=$(ls gpurun_out/r3/prof_te/*/*kernel_stats.csv | head -1) && head -30 $f | cut -c1-200
