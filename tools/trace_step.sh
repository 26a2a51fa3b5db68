#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of a few training steps under a list of OCT_OPTIONS settings, then the
# main stream's gap statistics (tools/trace_gaps.py).  Usage: tools/trace_step.sh <tag> "default" "name=v,..." ...
set -uo pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for O in "$@"; do
  i=$((i+1))
  if [ "$O" = "default" ]; then unset OCT_OPTIONS; else export OCT_OPTIONS="$O"; fi
  OUT=$ROOT/gpurun_out/trace_${TAG}_$i
  rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 5 --warmup 2 \
      --no-cpu-baseline --no-profile --no-inference --no-fit --no-collective-leg > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
  echo "== $O"
  python3 $ROOT/tools/trace_gaps.py $OUT | tee $OUT/gaps.txt
done
