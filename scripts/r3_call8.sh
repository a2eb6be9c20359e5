#!/bin/bash
O=gpurun_out/r3c8; mkdir -p $O
for v in product head; do
  L=build/lib_$v.so; [ $v = product ] && L=difflinker_amd/libdifflinker_hip.so
  DIFFLINKER_HIP_LIB=$L timeout 600 python bench.py --config C4 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | grep -o '"value": [0-9.]*' | sed "s/^/$v C4 /" >> $O/ab.log
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "pocket" 2>&1 | tail -n 2
cat $O/ab.log
