#!/bin/bash
O=gpurun_out/r3c7; mkdir -p $O
for rep in 1 2; do
for v in product head; do
  L=build/lib_$v.so; [ $v = product ] && L=difflinker_amd/libdifflinker_hip.so
  DIFFLINKER_HIP_LIB=$L timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | grep -o '"value": [0-9.]*' | sed "s/^/$v /" >> $O/ab.log
done; done
cat $O/ab.log
