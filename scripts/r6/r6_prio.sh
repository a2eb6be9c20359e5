#!/bin/bash
O=gpurun_out/r6/prio
mkdir -p $O
for cfg in "--batch 64 --team 4" "--batch 128 --team 2" "--batch 256 --team 1" "--batch 256 --team 1 --n 35"; do
  for lib in "" difflinker_amd/variants/lib_prio_static.so difflinker_amd/variants/lib_prio_strong.so; do
    DIFFLINKER_HIP_LIB=$lib timeout 300 python scripts/time_forward.py --raw --iters 50 $cfg 2>&1 | tail -1
  done
done | tee $O/forward.log
for lib in "" difflinker_amd/variants/lib_prio_static.so difflinker_amd/variants/lib_prio_strong.so ""; do
  echo "== lib: ${lib:-product}"
  DIFFLINKER_HIP_LIB=$lib timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('headline', round(d['value'],1), 'kernel_ms', round(d['roofline']['kernel_ms'],1), d.get('split_chain'))
"
done | tee $O/headline.log
