#!/bin/bash
# round 3, call 14: phase timelines of the team paths on the current build
O=gpurun_out/r3c14; mkdir -p $O
DIFFLINKER_HIP_LIB=build/lib_prof.so timeout 300 python scripts/phase_timeline.py --n 50 --batch 64 --team 4 > $O/tl_team4.log 2>&1
DIFFLINKER_HIP_LIB=build/lib_prof.so timeout 300 python scripts/phase_timeline.py --n 80 --batch 64 --team 2 > $O/tl_team2_n80.log 2>&1
head -36 $O/tl_team4.log
