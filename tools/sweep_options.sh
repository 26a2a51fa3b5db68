#!/bin/bash
# Runs on the GPU box: bench.py (train steps only) under a list of OCT_OPTIONS settings, one line each.
# Usage: tools/sweep_options.sh "name=v,name=v" "name=v" ...
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for O in "default" "$@"; do
  if [ "$O" = "default" ]; then unset OCT_OPTIONS; else export OCT_OPTIONS="$O"; fi
  R=$(timeout -k 10 120 python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile --no-inference --no-collective-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['step_ms_median_events'])") || R="failed"
  echo "$O : $R"
done
