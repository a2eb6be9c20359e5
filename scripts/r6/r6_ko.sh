#!/bin/bash
O=gpurun_out/r6/ko
mkdir -p $O
for cfg in "--batch 64 --team 1" "--batch 256 --team 1" "--batch 256 --team 1 --n 35"; do
  for lib in "" difflinker_amd/variants/lib_r5base.so difflinker_amd/variants/lib_ko_dma2.so difflinker_amd/variants/lib_ko_mfma2.so; do
    DIFFLINKER_HIP_LIB=$lib timeout 300 python scripts/time_forward.py --raw --iters 50 $cfg 2>&1 | tail -1
  done
done | tee $O/forward.log
