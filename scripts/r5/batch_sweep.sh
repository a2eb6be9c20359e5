#!/bin/bash
# round 5 (GPU box): C2 headline at several batch sizes, one JSON line each -> gpurun_out/r5/batch_sweep.log
OUT=gpurun_out/r5; mkdir -p $OUT
for b in "$@"; do
  python bench.py --batch $b --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B', d['config']['molecules_per_gpu'], 'mol/s %.1f' % d['value'], 'ms/chain %.1f' % d['ms_per_step'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'], 'exec_frac %.3f' % d['roofline']['frac'])"
done | tee $OUT/batch_sweep.log
