#!/bin/bash
# rocprofv3 kernel statistics of two secondary workloads (gpurun, from the repo root): C4 pockets and C2 at B = 64 on teams
set -u
OUT=gpurun_out/prof_secondary
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
run() {  # name, bench args...
  local name=$1; shift
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rp_$name --output-format csv -- python $ROOT/bench.py "$@" ) > $OUT/$name.log 2>&1
  echo "[$name] exit $?" >> $OUT/$name.log
  for f in $(find /tmp/rp_$name -name '*kernel_stats.csv' 2>/dev/null); do cp $f $OUT/${name}_kernel_stats.csv; done
  rm -rf /tmp/rp_$name
}
run c4_pockets --config C4 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary
run c2_b64_teams --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
ls -la $OUT
