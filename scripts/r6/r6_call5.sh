#!/bin/bash
O=gpurun_out/r6/call5
mkdir -p $O
for v in prof ko_dma ko_mfma ko_dsr; do
  DIFFLINKER_HIP_LIB=difflinker_amd/variants/lib_$v.so timeout 300 python scripts/phase_timeline.py --n 50 --batch 64 --team 1 > $O/phase_$v.log 2>&1; echo "$v exit $?"
  grep "^forward" $O/phase_$v.log
  grep "stream phase 1" $O/phase_$v.log
done
