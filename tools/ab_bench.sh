#!/bin/bash
# Runs on the GPU box: cfg-A (fp32) and cfg-C (bf16) train-step bench lines + launch tables under gpurun_out/$1_*
# usage: tools/ab_bench.sh <tag> [test-file ...]    (tests, when named, run first and gate the benches)
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd); TAG=$1; shift
mkdir -p $ROOT/gpurun_out
if [ $# -gt 0 ]; then
  timeout -k 10 900 python3 -m pytest "$@" -x -q -m gpu > $ROOT/gpurun_out/${TAG}_tests.log 2>&1 || { tail -30 $ROOT/gpurun_out/${TAG}_tests.log; exit 1; }
  tail -1 $ROOT/gpurun_out/${TAG}_tests.log
fi
python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --dump-profile $ROOT/gpurun_out/${TAG}_tableA.json > $ROOT/gpurun_out/${TAG}_A.json 2> $ROOT/gpurun_out/${TAG}_A.err || { tail -5 $ROOT/gpurun_out/${TAG}_A.err; exit 1; }
python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --batch 64 --height 512 --width 1024 --pool-layers 5 --act-dtype bf16 --dump-profile $ROOT/gpurun_out/${TAG}_tableC.json > $ROOT/gpurun_out/${TAG}_C.json 2> $ROOT/gpurun_out/${TAG}_C.err || { tail -5 $ROOT/gpurun_out/${TAG}_C.err; exit 1; }
python3 - <<PY
import json
for c in "AC":
    d = json.loads(open("$ROOT/gpurun_out/${TAG}_%s.json" % c).read().strip().splitlines()[-1])
    print(c, d["value"], d["ms_per_step"], d["step_ms_median_events"], {k: round(v, 3) for k, v in list(d["kernel_time_ms_per_step"].items())[:6]})
PY
