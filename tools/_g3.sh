set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
rm -rf gpurun_out/r2/prof_a gpurun_out/r2/prof_head gpurun_out/r2/pmc_mfma
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r2/a_unprofiled.json 2> gpurun_out/r2/a_unprofiled.err
tail -c 1500 gpurun_out/r2/a_unprofiled.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_a -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --repeats 1 > gpurun_out/r2/a_profiled.json 2> gpurun_out/r2/a_profiled.err
python3 bench.py --head-only --steps 200 --warmup 20 > gpurun_out/r2/head_unprofiled.json 2>> gpurun_out/r2/a_unprofiled.err
cat gpurun_out/r2/head_unprofiled.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof_head -- python3 bench.py --head-only --steps 50 --warmup 10 --repeats 1 > gpurun_out/r2/head_profiled.json 2> gpurun_out/r2/head_profiled.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r2/pmc_mfma -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --repeats 1 > gpurun_out/r2/pmc_mfma.json 2> gpurun_out/r2/pmc_mfma.err || (tail -5 gpurun_out/r2/pmc_mfma.err; rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|SQ_BUSY" | head -20)
ls gpurun_out/r2/prof_a/*/ gpurun_out/r2/pmc_mfma/*/ 2>/dev/null | head
du -sh gpurun_out/r2
