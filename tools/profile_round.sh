#!/bin/bash
# Regenerates the raw material of profiles/rNN_* on a GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <out_dir_under_gpurun_out>
# 1 unprofiled headline bench, 2 rocprofv3 kernel trace of the same command, 3-5 PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA; each in
# its own run with --kernel-trace only, as the MI355X guide prescribes), 6-7 head-only mode unprofiled + traced,
# 8-9 the data-parallel leg on one GPU (torchrun world 1, RCCL, UFND_FORCE_REDUCE=1: bucketed exchange live).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
rm -rf $O/prof $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/prof_head
python3 bench.py --steps 20 --warmup 5 > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-lookahead-compare --repeats 1 > $O/bench_profiled.json 2> $O/bench_profiled.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-lookahead-compare --repeats 1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-lookahead-compare --repeats 1 > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-lookahead-compare --repeats 1 > $O/pmc_mfma.json 2> $O/pmc_mfma.err
python3 bench.py --head-only --steps 200 --warmup 20 > $O/head_unprofiled.json 2> $O/head_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_head -- python3 bench.py --head-only --steps 50 --warmup 10 --repeats 1 > $O/head_profiled.json 2> $O/head_profiled.err
UFND_FORCE_REDUCE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --head-only --steps 200 --warmup 20 > $O/head_dp1.json 2> $O/head_dp1.err
UFND_FORCE_REDUCE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_dp1.json 2> $O/bench_dp1.err
for f in bench_unprofiled head_unprofiled head_dp1 bench_dp1; do python3 - <<PY
import json
d=json.loads([l for l in open("$O/$f.json") if l.startswith('{"metric"')][-1]); print("$f", d["value"], d["ms_per_step"])
PY
done
du -sh $O
