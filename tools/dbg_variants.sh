#!/bin/bash
# Runs on the GPU box: launch tables of cfg-A train steps under debug-variant libraries (timing experiments; results of
# those variants are WRONG by construction).  usage: tools/dbg_variants.sh <tag> lib1.so lib2.so ...
ROOT=$(cd "$(dirname "$0")/.." && pwd); TAG=$1; shift
mkdir -p $ROOT/gpurun_out
i=0
for L in default "$@"; do
  if [ "$L" = default ]; then unset OCT_UNET_LIB; else export OCT_UNET_LIB=$ROOT/$L; fi
  timeout -k 10 200 python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-collective-leg --dump-profile $ROOT/gpurun_out/${TAG}_table$i.json > $ROOT/gpurun_out/${TAG}_$i.json 2> $ROOT/gpurun_out/${TAG}_$i.err || tail -3 $ROOT/gpurun_out/${TAG}_$i.err
  python3 -c "
import json
d = json.loads(open('$ROOT/gpurun_out/${TAG}_$i.json').read().strip().splitlines()[-1])
print('$L', d['ms_per_step'], {k: round(v, 3) for k, v in list(d['kernel_time_ms_per_step'].items())[:5]})
"
  i=$((i+1))
done
