#!/bin/bash
# the final build of round 6 against the round-5 kernels (lib_r5base.so: round-5 sources; same Python layer, XCD order and hand-over), one box
O=gpurun_out/r6/vs_r5
V=difflinker_amd/variants/lib_r5base.so
mkdir -p $O
run() {
  python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'mol/s %.1f' % d['value'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'], d.get('split_chain'))"
}
for lib in "" $V "" $V "" $V; do DIFFLINKER_HIP_LIB=$lib run "lib ${lib:-product}"; done | tee $O/ab.log
for cfg in "--batch 256 --team 1 --n 35" "--batch 256 --team 1 --n 36" "--batch 256 --team 1 --n 44" "--batch 256 --team 1 --n 50" "--batch 64 --team 4" "--batch 128 --team 2"; do
  for lib in "" $V; do
    DIFFLINKER_HIP_LIB=$lib timeout 300 python scripts/time_forward.py --raw --iters 50 $cfg 2>&1 | tail -1
  done
done | tee $O/forward.log
