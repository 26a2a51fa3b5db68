#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel-trace stats + separate PMC passes (HBM traffic).
# Usage: tools/profile_round.sh <tag> [extra bench args]      outputs under gpurun_out/prof_<tag>/
# The PMC passes run TRAIN STEPS ONLY (--no-inference), so the sum over all kernels / steps is the step's HBM traffic.
set -uo pipefail
TAG=${1:-r02}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 20 --warmup 5 --dump-profile $OUT/launch_table.json "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
echo "bench: $(python3 -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['unit'],d['ms_per_step'],'ms/step')")"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile "$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
echo "kernel trace done"
PMC_STEPS=2
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 $ROOT/bench.py --steps $PMC_STEPS --warmup 0 --no-cpu-baseline --no-profile --no-inference "$@" > $OUT/pmc_$C.log 2>&1 || { tail -5 $OUT/pmc_$C.log; exit 1; }
  echo "pmc $C done"
done
echo $PMC_STEPS > $OUT/pmc_steps.txt
# keep only what is needed (the raw per-dispatch traces are large)
find $OUT -name '*kernel_trace.csv' -size +20M -delete
ls $OUT
