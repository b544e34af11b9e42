#!/bin/bash
# Regenerates the raw material of profiles/rNN_* on a GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <out_dir_under_gpurun_out> [part ...]      parts: main pmc head dp other train (default: all)
# main  1 unprofiled headline bench, 2 rocprofv3 kernel trace of the same command
# pmc   3-5 PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA; each in its own run with --kernel-trace only, as the MI355X guide prescribes)
# head  head-only mode unprofiled + traced (B = 32 and B = 256)
# dp    the data-parallel leg on one GPU (torchrun world 1, RCCL, --force-exchange: bucketed exchange live)
# other BASELINE configs[3] (L = 512, B = 128) and configs[4]'s per-GPU shard (8 frames, B = 8): unprofiled + traced
# train bench --train-encoders unprofiled + traced
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1
shift
PARTS="${@:-main pmc head dp other train}"
mkdir -p $O
has() { [[ " $PARTS " == *" $1 "* ]]; }
RP="rocprofv3 --kernel-trace --stats --output-format csv"
Q="--no-cpu-baseline --no-lookahead-compare --repeats 1"
if has main; then
  rm -rf $O/prof
  python3 bench.py --steps 20 --warmup 5 > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
  $RP -d $O/prof -- python3 bench.py --steps 20 --warmup 5 $Q > $O/bench_profiled.json 2> $O/bench_profiled.err
fi
if has pmc; then
  rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 4 --warmup 2 $Q > $O/pmc_fetch.json 2> $O/pmc_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 4 --warmup 2 $Q > $O/pmc_write.json 2> $O/pmc_write.err
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 4 --warmup 2 $Q > $O/pmc_mfma.json 2> $O/pmc_mfma.err
fi
if has head; then
  rm -rf $O/prof_head $O/prof_head256
  python3 bench.py --head-only --steps 200 --warmup 20 > $O/head_unprofiled.json 2> $O/head_unprofiled.err
  $RP -d $O/prof_head -- python3 bench.py --head-only --steps 50 --warmup 10 --repeats 1 > $O/head_profiled.json 2> $O/head_profiled.err
  python3 bench.py --head-only --batch 256 --steps 100 --warmup 20 > $O/head256_unprofiled.json 2> $O/head256_unprofiled.err
  $RP -d $O/prof_head256 -- python3 bench.py --head-only --batch 256 --steps 30 --warmup 10 --repeats 1 > $O/head256_profiled.json 2> $O/head256_profiled.err
fi
if has dp; then
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --head-only --force-exchange --steps 200 --warmup 20 > $O/head_dp1.json 2> $O/head_dp1.err
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --force-exchange --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_dp1.json 2> $O/bench_dp1.err
fi
if has other; then
  rm -rf $O/prof_L512 $O/prof_F8
  python3 bench.py --seq-len 512 --batch 128 --steps 8 --warmup 4 --repeats 3 --no-cpu-baseline > $O/L512_unprofiled.json 2> $O/L512_unprofiled.err
  $RP -d $O/prof_L512 -- python3 bench.py --seq-len 512 --batch 128 --steps 8 --warmup 4 $Q > $O/L512_profiled.json 2> $O/L512_profiled.err
  python3 bench.py --frames 8 --batch 8 --steps 40 --warmup 8 --repeats 3 --no-cpu-baseline > $O/F8_unprofiled.json 2> $O/F8_unprofiled.err
  $RP -d $O/prof_F8 -- python3 bench.py --frames 8 --batch 8 --steps 20 --warmup 8 $Q > $O/F8_profiled.json 2> $O/F8_profiled.err
fi
if has train; then
  rm -rf $O/prof_train
  python3 bench.py --train-encoders --steps 10 --warmup 3 --repeats 3 > $O/train_unprofiled.json 2> $O/train_unprofiled.err
  $RP -d $O/prof_train -- python3 bench.py --train-encoders --steps 6 --warmup 2 --repeats 1 > $O/train_profiled.json 2> $O/train_profiled.err
fi
for f in bench_unprofiled head_unprofiled head256_unprofiled head_dp1 bench_dp1 L512_unprofiled F8_unprofiled train_unprofiled; do [ -f $O/$f.json ] && python3 - <<PY
import json
d=json.loads([l for l in open("$O/$f.json") if l.startswith('{"metric"')][-1]); print("$f", d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"))
PY
done
du -sh $O
