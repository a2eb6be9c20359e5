#!/bin/bash
# round 6: A/B on teams and small molecules (M-split of the per-atom GEMM chain)
O=gpurun_out/r6/ab_$1
mkdir -p $O
B=${2:-difflinker_amd/variants/lib_r5base.so}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_team.py tests/test_gpu_cabi.py -q -x 2>&1 | tail -n 2
for cfg in "--batch 64 --team 4" "--batch 128 --team 2" "--batch 64 --team 1 --n 30" "--batch 256 --team 1 --n 16" "--batch 64 --team 4 --n 36" "--batch 256 --team 1"; do
  for lib in "" $B; do
    DIFFLINKER_HIP_LIB=$lib timeout 300 python scripts/time_forward.py --raw --iters 50 $cfg 2>&1 | tail -1
  done
done | tee $O/forward.log
for b in 64 128; do
for lib in "" $B; do
  echo "== B=$b lib: ${lib:-product}"
  DIFFLINKER_HIP_LIB=$lib timeout 600 python bench.py --batch $b --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('molecules/s', round(d['value'],1), 'kernel_ms', round(d['roofline']['kernel_ms'],1))
"
done; done | tee $O/batch.log
