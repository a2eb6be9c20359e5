#!/bin/bash
O=gpurun_out/r6/call4
mkdir -p $O
export DIFFLINKER_HIP_LIB=difflinker_amd/variants/lib_prof.so
timeout 300 python scripts/phase_timeline.py --n 50 --batch 64 --team 1 > $O/phase_B64_n50_team1.log 2>&1; echo "exit $?"
timeout 300 python scripts/phase_timeline.py --n 50 --batch 256 --team 1 > $O/phase_B256_n50.log 2>&1; echo "exit $?"
grep "stream phase" $O/phase_B64_n50_team1.log
echo; grep "stream phase" $O/phase_B256_n50.log
