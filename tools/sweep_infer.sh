#!/bin/bash
# Runs on the GPU box: hipGraph inference forward (batch 128) ms per B-scan under a list of OCT_OPTIONS settings.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for O in "default" "$@"; do
  if [ "$O" = "default" ]; then unset OCT_OPTIONS; else export OCT_OPTIONS="$O"; fi
  R=$(OCT_BENCH_NO_E2E=1 timeout -k 10 120 python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-fit --no-collective-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['inference_ms_per_scan'])") || R="failed"
  echo "$O : $R"
done
