#!/bin/bash
O=gpurun_out/r6/fuzz
mkdir -p $O
{
for lib in "" difflinker_amd/variants/lib_r5base.so; do
  echo "== lib ${lib:-product}"
  DIFFLINKER_HIP_LIB=$lib timeout 600 python scripts/r5/fuzz_forward.py --seed 6 --only 77 2>&1 | tail -n 3
  DIFFLINKER_HIP_LIB=$lib timeout 600 python scripts/r5/fuzz_forward.py --seed 6 --only 77 --set precision=\'fp32\' 2>&1 | tail -n 2
  DIFFLINKER_HIP_LIB=$lib timeout 600 python scripts/r5/fuzz_forward.py --seed 6 --only 77 --set "sizes=(31, 31)" 2>&1 | tail -n 2
  DIFFLINKER_HIP_LIB=$lib timeout 600 python scripts/r5/fuzz_forward.py --seed 6 --only 77 --set "sizes=(55, 55)" 2>&1 | tail -n 2
  DIFFLINKER_HIP_LIB=$lib timeout 600 python scripts/r5/fuzz_forward.py --seed 6 --only 77 --set mag=1.0 2>&1 | tail -n 2
done
} > $O/replay_600077.log 2>&1
cat $O/replay_600077.log
