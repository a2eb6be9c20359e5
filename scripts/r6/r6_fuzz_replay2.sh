#!/bin/bash
O=gpurun_out/r6/fuzz
mkdir -p $O
{
for ov in "aggregation_method='sum'" "trained=False" "mag=100.0" "mag=30.0" "mag=10.0" "sub=1" "ctx=1" "nf=8" "hidden=64" "ct=False" "L=2"; do
  echo "-- override $ov"
  timeout 600 python scripts/r5/fuzz_forward.py --seed 6 --only 77 --set "$ov" 2>&1 | grep "case 77" | head -1
done
} > $O/replay2_600077.log 2>&1
cat $O/replay2_600077.log
