#!/bin/bash
# round 3, call 24: pocket node kernel with fragment rows (nodev2) against the product: C4 throughput, bitwise comparison of a forward, tests
mkdir -p gpurun_out/r3c24
for lib in prod nodev2 prod nodev2; do
  DIFFLINKER_HIP_LIB=build/lib_$lib.so timeout 600 python bench.py --config C4 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib C4', round(d['value'],2), round(d['ms_per_step'],1))" | tee -a gpurun_out/r3c24/ab.log
done
DIFFLINKER_HIP_LIB=build/lib_nodev2.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_hard.py tests/test_gpu_flags.py tests/test_gpu_generate.py -x -q -m gpu -k "pocket or large or sin" 2>&1 | tail -2 | tee -a gpurun_out/r3c24/ab.log
