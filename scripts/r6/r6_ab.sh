#!/bin/bash
# A/B of the product library against a variant (default: the round-5 per-atom phases) on ONE box: uniform forwards, the headline chain
O=gpurun_out/r6/ab_$1
mkdir -p $O
B=${2:-difflinker_amd/variants/lib_r5base.so}
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "forward_vs_oracle or golden" 2>&1 | tail -n 2
for cfg in "--batch 64 --team 1" "--batch 64 --team 4" "--batch 256 --team 1" "--batch 256 --team 1 --n 35" "--batch 256 --team 1 --n 44"; do
  for lib in "" $B; do
    DIFFLINKER_HIP_LIB=$lib timeout 300 python scripts/time_forward.py --raw --iters 50 $cfg 2>&1 | tail -1
  done
done | tee $O/forward.log
for lib in "" $B "" $B; do
  echo "== lib: ${lib:-product}"
  DIFFLINKER_HIP_LIB=$lib timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('headline', round(d['value'],1), 'kernel_ms', round(d['roofline']['kernel_ms'],1), d.get('split_chain'))
"
done | tee $O/headline.log
DIFFLINKER_HIP_LIB=difflinker_amd/variants/lib_prof.so timeout 300 python scripts/phase_timeline.py --n 50 --batch 64 --team 1 > $O/phase_B64.log 2>&1
grep "wave 0 stream phase\|wave 4 stream phase 1" $O/phase_B64.log
