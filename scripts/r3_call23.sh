#!/bin/bash
# round 3, call 23: A/B - operands of a step prefetched under the previous epilogue (xpf) against the product
mkdir -p gpurun_out/r3c23
for lib in prod xpf0 xpf1 prod xpf0 xpf1; do
  DIFFLINKER_HIP_LIB=build/lib_$lib.so timeout 300 python scripts/time_forward.py --batch 64 --team 1 2>/dev/null | tail -1 | tee -a gpurun_out/r3c23/ab.log
  DIFFLINKER_HIP_LIB=build/lib_$lib.so timeout 600 python bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib bench C2', round(d['value'],1), round(d['ms_per_step'],1))" | tee -a gpurun_out/r3c23/ab.log
done
