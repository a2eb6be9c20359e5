#!/bin/bash
# slot plans that take one / two more steps where that takes fewer wave-steps, against the fewest-slots plan (product); uniform forwards
O=gpurun_out/r6/plan
mkdir -p $O
for cfg in "--batch 256 --team 1 --n 38" "--batch 256 --team 1 --n 40" "--batch 256 --team 1 --n 47" "--batch 256 --team 1 --n 48" "--batch 256 --team 1 --n 50" "--batch 64 --team 4"; do
  for lib in "" difflinker_amd/variants/lib_extra1.so difflinker_amd/variants/lib_extra2.so; do
    DIFFLINKER_HIP_LIB=$lib timeout 300 python scripts/time_forward.py --raw --iters 50 $cfg 2>&1 | tail -1
  done
done | tee $O/forward.log
