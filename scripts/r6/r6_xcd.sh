#!/bin/bash
# round 6: size-sorted molecules dealt to the XCDs in contiguous runs (DIFFLINKER_XCD_ORDER): speed and fabric traffic, one box
O=gpurun_out/r6/xcd
mkdir -p $O
run() {
  python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'mol/s %.1f' % d['value'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'], 'traffic GB/chain %.1f' % ((d['roofline']['traffic'] or 0)/1e9), d.get('split_chain'))"
}
for x in 2 4 0 2 4 0; do DIFFLINKER_XCD_ORDER=$x run "xcd_order $x"; done | tee $O/ab_xcd_order.log
