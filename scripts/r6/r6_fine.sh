#!/bin/bash
# A/B: pair-loop progress published 2 / 4 times per step (finer balancing of the two waves of a SIMD) against the product
O=gpurun_out/r6/fine
mkdir -p $O
run() {
  python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'mol/s %.1f' % d['value'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'])"
}
for rep in 1 2; do for lib in "" difflinker_amd/variants/lib_fine2.so difflinker_amd/variants/lib_fine4.so; do DIFFLINKER_HIP_LIB=$lib run "lib ${lib:-product}"; done; done | tee $O/ab.log
for cfg in "--batch 256 --team 1 --n 35" "--batch 256 --team 1" "--batch 64 --team 4" "--batch 128 --team 2"; do
  for lib in "" difflinker_amd/variants/lib_fine2.so difflinker_amd/variants/lib_fine4.so; do
    DIFFLINKER_HIP_LIB=$lib timeout 300 python scripts/time_forward.py --raw --iters 50 $cfg 2>&1 | tail -1
  done
done | tee $O/forward.log
