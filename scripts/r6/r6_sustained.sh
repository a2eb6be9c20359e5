#!/bin/bash
# the driver's command line (--steps 20 --warmup 5) against the short default (3 / 1) on one box: does the rate hold over 25 chains?
O=gpurun_out/r6/sustained
mkdir -p $O
run() {
  python bench.py $2 --no-secondary --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'mol/s %.1f' % d['value'], 'ms/step %.1f' % d['ms_per_step'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'], 'frac %.4f' % d['roofline']['frac'])"
}
{
run "steps 3 warmup 1  " "--steps 3 --warmup 1"
run "steps 20 warmup 5 " "--gpus 1 --steps 20 --warmup 5"
run "steps 3 warmup 1  " "--steps 3 --warmup 1"
run "steps 20 warmup 5 " "--gpus 1 --steps 20 --warmup 5"
} | tee $O/ab.log
