#!/bin/bash
# Turns the raw material of one tools/profile_round.sh run (gpurun_out/<dir>) into profiles/<tag>_* (run from the repo root):
#   bash tools/profiles_from_round.sh gpurun_out/r3/final r03
set -e
P=$1; T=$2
Q="--no-cpu-baseline --no-lookahead-compare --repeats 1"
python3 tools/profile_summary.py $P/prof $P/bench_profiled.json ${T}_bench 28 $P/bench_unprofiled.json > /dev/null
python3 tools/pmc_summary.py $P/pmc_fetch $P/pmc_write profiles/${T}_pmc_traffic.md > /dev/null
python3 tools/pmc_mfma_summary.py $P/pmc_mfma profiles/${T}_pmc_mfma.md > /dev/null
python3 tools/head_profile_summary.py $P/prof_head $P/head_profiled.json $P/head_unprofiled.json ${T}_head 50 > /dev/null
python3 tools/head_profile_summary.py $P/prof_head256 $P/head256_profiled.json $P/head256_unprofiled.json ${T}_head256 30 > /dev/null
python3 tools/profile_summary.py $P/prof_L512 $P/L512_profiled.json ${T}_L512 9 $P/L512_unprofiled.json "--seq-len 512 --batch 128 --steps 8 --warmup 4 $Q" > /dev/null
python3 tools/profile_summary.py $P/prof_F8 $P/F8_profiled.json ${T}_F8 12 $P/F8_unprofiled.json "--frames 8 --batch 8 --steps 20 --warmup 8 $Q" > /dev/null
python3 tools/profile_summary.py $P/prof_train $P/train_profiled.json ${T}_train 8 $P/train_unprofiled.json "--train-encoders --steps 6 --warmup 2 --repeats 1" > /dev/null
ls profiles/${T}_*
