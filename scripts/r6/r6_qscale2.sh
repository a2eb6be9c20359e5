#!/bin/bash
# the hand-over call of split_plan scaled (DIFFLINKER_SPLIT_QSCALE), final build of round 6, one box
O=gpurun_out/r6/qscale2
mkdir -p $O
run() {
  python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'mol/s %.1f' % d['value'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'], d.get('split_chain'))"
}
for q in 1.0 0.96 0.98 1.02 1.04 1.06 1.0; do DIFFLINKER_SPLIT_QSCALE=$q run "qscale $q"; done | tee $O/ab_split_qscale.log
