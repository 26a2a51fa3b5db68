#!/bin/bash
# sweep launch-geometry options; prints scans/s per setting
for o in "" "dw32_blocks=384" "dw32_blocks=768" "dw32_blocks=1024" "dw16_blocks=512" "dw16_blocks=1024" "dw16_blocks=1536" "igemm_persistent_blocks=1024" "igemm_persistent_blocks=1536" "igemm_persistent_blocks=2048" "igemm_min_blocks=256" "igemm_min_blocks=768" "igemm_min_blocks=1024"; do
  v=$(OCT_OPTIONS=$o python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['inference_ms_per_scan'])")
  echo "[$o] $v"
done
