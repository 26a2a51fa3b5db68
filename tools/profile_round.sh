#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel-trace stats + separate PMC passes (HBM traffic).
# Usage: tools/profile_round.sh <tag> [extra bench args]      outputs under gpurun_out/prof_<tag>/
# The PMC passes run TRAIN STEPS ONLY (--no-inference), so the sum over all kernels / steps is the step's HBM traffic.
set -uo pipefail
TAG=${1:-r02}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT            # (gpurun merges gpurun_out/ back: without this, older runs' traces pile up beside the new ones)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --steps 20 --warmup 5 --dump-profile $OUT/launch_table.json "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
echo "bench: $(python3 -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['unit'],d['ms_per_step'],'ms/step')")"
# every profiler pass runs TRAIN STEPS ONLY, no host-side legs (no worker pools under the profiler's preload, no second
# engine): the per-kernel averages of the trace are then in-step durations of the B = 32 training launches alone
LEAN="--no-cpu-baseline --no-profile --no-inference --no-fit --no-collective-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 5 --warmup 2 $LEAN "$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
echo "kernel trace (training) done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_infer -- python3 $ROOT/tools/profile_infer.py > $OUT/trace_infer.log 2>&1 || { tail -5 $OUT/trace_infer.log; echo "(inference trace failed: continuing)"; }
PMC_STEPS=2
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 $ROOT/bench.py --steps $PMC_STEPS --warmup 0 $LEAN "$@" > $OUT/pmc_$C.log 2>&1 || { tail -5 $OUT/pmc_$C.log; exit 1; }
  echo "pmc $C done"
done
echo $PMC_STEPS > $OUT/pmc_steps.txt
# SQ counters (8 slots per pass): matrix-pipe busy cycles, instruction mix, wait states, LDS conflicts
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_pass$i -- python3 $ROOT/bench.py --steps 1 --warmup 1 $LEAN "$@" > $OUT/sq_pass$i.log 2>&1 || { tail -5 $OUT/sq_pass$i.log; echo "(SQ pass $i failed: continuing)"; }
  echo "sq pass $i done"
done
# keep only what is needed (the raw per-dispatch traces are large)
find $OUT -name '*kernel_trace.csv' -size +20M -delete
ls $OUT
