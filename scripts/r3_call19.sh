#!/bin/bash
# round 3, call 19: A/B - last k-slab of the pair loop tile-major with the first epilogues in its shadow (tail) vs product
mkdir -p gpurun_out/r3c19
for lib in prod tail prod tail; do
  for a in "--batch 64 --team 1" "--batch 256 --team 1"; do
    echo "== $lib $a" | tee -a gpurun_out/r3c19/ab.log
    DIFFLINKER_HIP_LIB=build/lib_$lib.so timeout 300 python scripts/time_forward.py $a 2>/dev/null | tail -1 | tee -a gpurun_out/r3c19/ab.log
  done
  DIFFLINKER_HIP_LIB=build/lib_$lib.so timeout 600 python bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench C2', round(d['value'],1), round(d['ms_per_step'],1))" | tee -a gpurun_out/r3c19/ab.log
done
DIFFLINKER_HIP_LIB=build/lib_tail.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
