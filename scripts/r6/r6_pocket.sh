#!/bin/bash
O=gpurun_out/r6/pocket_$1
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -k "pocket or large or sin or flags or xl or C4" 2>&1 | tail -n 2
for lib in "" difflinker_amd/variants/lib_r5base.so; do
  echo "== lib: ${lib:-product}"
  DIFFLINKER_HIP_LIB=$lib timeout 900 python bench.py --config C4 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C4 molecules/s', round(d['value'],2), 'ms/chain', round(d['ms_per_step'],1))
"
done | tee $O/c4.log
