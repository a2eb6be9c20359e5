#!/bin/bash
# round 3, call 15: A/B - scratch tiles moved with 16-byte accesses (tile4) against the previous commit (both)
mkdir -p gpurun_out/r3c15
for lib in u1skip parts u1skip parts; do
  for a in "--batch 64 --team auto" "--batch 128 --team 2" "--batch 256 --team 1"; do
    echo "== $lib $a" | tee -a gpurun_out/r3c15/ab.log
    DIFFLINKER_HIP_LIB=build/lib_$lib.so timeout 300 python scripts/time_forward.py $a 2>/dev/null | tail -1 | tee -a gpurun_out/r3c15/ab.log
  done
done
DIFFLINKER_HIP_LIB=build/lib_parts.so timeout 900 python -m pytest tests/test_gpu_team.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
