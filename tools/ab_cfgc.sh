#!/bin/bash
# Runs on the GPU box: BASELINE configs[2] (512x1024, P=5, B=64, bf16) train-step bench under each OCT_OPTIONS setting.
ROOT=$(cd "$(dirname "$0")/.." && pwd); TAG=$1; shift
i=0
for O in "$@"; do
  if [ "$O" = "default" ]; then unset OCT_OPTIONS; else export OCT_OPTIONS="$O"; fi
  timeout -k 10 300 python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-collective-leg --batch 64 --height 512 --width 1024 --pool-layers 5 --act-dtype bf16 --dump-profile $ROOT/gpurun_out/${TAG}_tableC$i.json > $ROOT/gpurun_out/${TAG}_C$i.json 2> $ROOT/gpurun_out/${TAG}_C$i.err || tail -3 $ROOT/gpurun_out/${TAG}_C$i.err
  python3 -c "
import json
d = json.loads(open('$ROOT/gpurun_out/${TAG}_C$i.json').read().strip().splitlines()[-1])
print('cfgC $O', d['value'], d['ms_per_step'], {k: round(v, 3) for k, v in list(d['kernel_time_ms_per_step'].items())[:7]})
"
  i=$((i+1))
done
