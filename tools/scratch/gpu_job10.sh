#!/bin/bash
# tile / lookahead re-test after the spill fix
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/x10; mkdir -p $O
Q="--no-cpu-baseline --no-lookahead-compare --repeats 3 --steps 40 --warmup 8"
run() { n=$1; shift; python3 bench.py $Q "$@" > $O/$n.json 2> $O/$n.err; python3 - <<PY
import json
d=json.loads([l for l in open("$O/$n.json") if l.startswith('{"metric"')][-1]); print("$n", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
}
run base
run t_out15 --text-tiles out=15,ffn2=15
run t_ffn2_15 --text-tiles ffn2=15
run v_15 --vis-tiles out=15,ffn2=15
run v_ffn1_15 --vis-tiles ffn1=15
run la8 --lookahead 8
run base2
