#!/bin/bash
# round 3, call 11: kernel statistics of the secondary workloads (C4 pockets, C2 at B = 64 on teams, C2L) on the current build
bash scripts/profile_secondary.sh > /dev/null 2>&1
OUT=gpurun_out/prof_secondary
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_c2l --output-format csv -- python /root/repo/bench.py --config C2L --steps 1 --warmup 1 --no-cpu-baseline --no-secondary ) > $OUT/c2l.log 2>&1
for f in $(find /tmp/rp_c2l -name '*kernel_stats.csv' 2>/dev/null); do cp $f $OUT/c2l_kernel_stats.csv; done
for f in $OUT/*_kernel_stats.csv; do echo "== $f"; head -12 $f | cut -c1-200; done
tail -1 $OUT/c4_pockets.log | cut -c1-300
