#!/bin/bash
O=gpurun_out/r6/call3
mkdir -p $O
export DIFFLINKER_HIP_LIB=difflinker_amd/variants/lib_prof.so
timeout 300 python scripts/phase_timeline.py --n 50 --batch 64 --team 1 > $O/phase_B64_n50_team1.log 2>&1; echo "exit $?"
timeout 300 python scripts/phase_timeline.py --n 50 --batch 256 --team 1 > $O/phase_B256_n50.log 2>&1; echo "exit $?"
timeout 300 python scripts/phase_timeline.py --n 50 --batch 64 --team 4 > $O/phase_B64_team4.log 2>&1; echo "exit $?"
unset DIFFLINKER_HIP_LIB
for cfg in "--batch 64 --team 1" "--batch 64 --team 4" "--batch 256 --team 1" "--batch 256 --team 1 --n 35"; do
  for lib in "" difflinker_amd/variants/lib_r5base.so; do
    DIFFLINKER_HIP_LIB=$lib timeout 300 python scripts/time_forward.py $cfg 2>&1 | tail -1
  done
done
