#!/bin/bash
# round 3, call 5: machine-scheduler variants on the round-3 build; team timeline
O=gpurun_out/r3c5; mkdir -p $O
for v in product sched_default sched_maxilp sched_maxmem sched_iterminreg; do
  L=build/lib_$v.so; [ $v = product ] && L=difflinker_amd/libdifflinker_hip.so
  for b in 64 256; do DIFFLINKER_HIP_LIB=$L timeout 200 python scripts/time_forward.py --batch $b --team 1 2>&1 | tail -n 1 >> $O/tf_$v.log; done
  DIFFLINKER_HIP_LIB=$L timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | grep -o '"value": [0-9.]*' >> $O/tf_$v.log
done
DIFFLINKER_HIP_LIB=build/lib_prof_v2.so timeout 300 python scripts/phase_timeline.py --n 50 --batch 64 --team 4 > $O/tl_team4.log 2>&1
DIFFLINKER_HIP_LIB=build/lib_prof_v2.so timeout 300 python scripts/phase_timeline.py --n 50 --batch 64 --team 1 > $O/tl_team1.log 2>&1
tail -n 3 $O/tf_*.log
