#!/bin/bash
# round 5 (GPU box): where the big molecules of the split chain should stop - the planner's call scaled by a factor -> gpurun_out/r5/ab_split_qscale.log
O=gpurun_out/r5; mkdir -p $O
run() {
  python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'mol/s %.1f' % d['value'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'], d.get('split_chain'))"
}
for q in 1.0 0.94 0.97 1.03 1.06 1.0; do DIFFLINKER_SPLIT_QSCALE=$q run "qscale $q"; done | tee $O/ab_split_qscale.log
