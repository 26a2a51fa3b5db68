#!/bin/bash
# Runs on the GPU box: cfg-A train-step bench + per-(kernel, layer) launch table under each OCT_OPTIONS setting.
# usage: tools/ab_tables.sh <tag> "default" "name=v,name=v" ...   -> gpurun_out/<tag>_<i>.json / _table<i>.json
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd); TAG=$1; shift
mkdir -p $ROOT/gpurun_out
i=0
for O in "$@"; do
  if [ "$O" = "default" ]; then unset OCT_OPTIONS; else export OCT_OPTIONS="$O"; fi
  timeout -k 10 200 python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-collective-leg --dump-profile $ROOT/gpurun_out/${TAG}_table$i.json > $ROOT/gpurun_out/${TAG}_$i.json 2> $ROOT/gpurun_out/${TAG}_$i.err || { tail -5 $ROOT/gpurun_out/${TAG}_$i.err; }
  python3 -c "
import json
d = json.loads(open('$ROOT/gpurun_out/${TAG}_$i.json').read().strip().splitlines()[-1])
print('$O', d['value'], d['ms_per_step'], d['step_ms_median_events'], {k: round(v, 3) for k, v in list(d['kernel_time_ms_per_step'].items())[:8]})
"
  i=$((i+1))
done
