#!/bin/bash
# round 6: the hand-over in K stages (EDM.split_stages / DIFFLINKER_SPLIT_STAGES): the headline by K, one box
O=gpurun_out/r6/stages
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round5.py -q -x -k "split_chain" 2>&1 | tail -n 1
run() {
  python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'mol/s %.1f' % d['value'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'], d.get('split_chain'))"
}
for k in 2 3 4 6 8 2 6; do DIFFLINKER_SPLIT_STAGES=$k run "stages $k"; done | tee $O/ab_split_stages.log
DIFFLINKER_SPLIT_CHAIN=0 run "one launch" | tee -a $O/ab_split_stages.log
