#!/bin/bash
# Runs on the GPU box: several rocprofv3 --pmc passes (counters only + kernel trace) over a short bench run.
# (TA_* / TCP_* counters: at most 2 per pass -- more fails with "Request exceeds the capabilities of the hardware")
# Usage: tools/pmc_passes.sh <outtag> "<CTRS pass 1>" "<CTRS pass 2>" ...    -> gpurun_out/pmc_<outtag>/pass<i>/
set -uo pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  # hard per-pass limit: a counter set the hardware cannot collect makes rocprofv3 abort and then hang in its finalizer
  timeout -k 10 ${PASS_TIMEOUT:-180} rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --infer-batch 32 ${BENCH_EXTRA:-} > $OUT/pass$i.log 2>&1 || { tail -5 $OUT/pass$i.log; exit 1; }
  echo "pass $i done: $C"
  find $OUT/pass$i -name '*kernel_trace.csv' -delete
done
