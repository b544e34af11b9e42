set -o pipefail
mkdir -p gpurun_out/r3
Q="--steps 40 --warmup 8 --no-cpu-baseline --no-lookahead-compare --repeats 3"
for cfg in "" "--text-tiles out=15,ffn2=15" "--text-tiles ffn2=15" "" "--text-tiles out=15,ffn2=15" "--lookahead 8"; do
  echo "cfg: $cfg" >> gpurun_out/r3/tiles_ab.log
  timeout -k 10 200 python bench.py $Q $cfg >> gpurun_out/r3/tiles_ab.log 2>gpurun_out/r3/tiles_ab.err || echo fail >> gpurun_out/r3/tiles_ab.log
done
python - <<'PY'
import json
for l in open("gpurun_out/r3/tiles_ab.log"):
    if l.startswith("cfg"): print(l.strip())
    elif l.startswith("{"):
        d=json.loads(l); s=d["roofline"]["per_launch"]["by_shape_MxNxK"]
        print("   ", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k: v["avg_us"] for k, v in s.items() if k.startswith(("16384x768", "32768x768"))})
PY
