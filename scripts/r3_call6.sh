#!/bin/bash
O=gpurun_out/r3c6; mkdir -p $O
for v in product dma_after pre_after; do
  L=build/lib_$v.so; [ $v = product ] && L=difflinker_amd/libdifflinker_hip.so
  for b in 64 256; do DIFFLINKER_HIP_LIB=$L timeout 200 python scripts/time_forward.py --batch $b --team 1 2>&1 | tail -n 1 >> $O/tf_$v.log; done
  DIFFLINKER_HIP_LIB=$L timeout 200 python scripts/time_forward.py --batch 64 --team 4 2>&1 | tail -n 1 >> $O/tf_$v.log
  DIFFLINKER_HIP_LIB=$L timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | grep -o '"value": [0-9.]*' >> $O/tf_$v.log
done
DIFFLINKER_HIP_LIB=difflinker_amd/libdifflinker_hip.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "forward_vs_oracle or chain_vs_oracle" 2>&1 | tail -n 2
tail -n 4 $O/tf_*.log
