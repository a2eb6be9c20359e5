#!/bin/bash
# round 5 (GPU box): same-box A/B of the split chain (static hand-over) on the C2 headline -> gpurun_out/r5/ab_split.log
O=gpurun_out/r5; mkdir -p $O
run() {
  python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'mol/s %.1f' % d['value'], 'ms/chain %.1f' % d['ms_per_step'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'], 'exec_frac %.4f' % d['roofline']['frac'], d.get('split_chain'))"
}
for rep in 1 2; do
  DIFFLINKER_SPLIT_CHAIN=0 run "one launch            "
  DIFFLINKER_SPLIT_CHAIN=1 DIFFLINKER_SPLIT_SINGLES=0 run "split, teams only     "
  DIFFLINKER_SPLIT_CHAIN=1 DIFFLINKER_SPLIT_SINGLES=1 run "split, teams + singles"
done | tee $O/ab_split.log
