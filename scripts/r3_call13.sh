#!/bin/bash
# round 3, call 13: A/B on one box - base (HEAD) vs exchange-only vs exchange + chunked reduction
mkdir -p gpurun_out/r3c13
for lib in base exonly both base; do
  for a in "--batch 64 --team auto" "--batch 128 --team 2"; do
    echo "== $lib $a" | tee -a gpurun_out/r3c13/ab.log
    DIFFLINKER_HIP_LIB=build/lib_$lib.so timeout 300 python scripts/time_forward.py $a 2>/dev/null | tail -1 | tee -a gpurun_out/r3c13/ab.log
  done
  DIFFLINKER_HIP_LIB=build/lib_$lib.so timeout 600 python bench.py --batch 64 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench B=64', round(d['value'],1), round(d['ms_per_step'],1))" | tee -a gpurun_out/r3c13/ab.log
done
DIFFLINKER_HIP_LIB=build/lib_both.so timeout 900 python -m pytest tests/test_gpu_team.py -x -q -m gpu 2>&1 | tail -2
