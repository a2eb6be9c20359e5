#!/bin/bash
# round 5 (GPU box, through gpurun from the repo root): PC sampling of the forward kernel on a uniform batch.
#   scripts/r5/pc_sample.sh [LIB]   ->  gpurun_out/r5/pc_*.{log,csv}
# LIB: a build with line tables (scripts/build_variant.sh lines -gline-tables-only) so that the samples carry file:line.
set -u
LIB=${1:-build/lib_lines.so}
OUT=gpurun_out/r5
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
( rocprofv3-avail list --pc-sampling; rocprofv3-avail info --pc-sampling ) > $OUT/pc_avail.log 2>&1
cat $OUT/pc_avail.log | head -60
export DIFFLINKER_HIP_LIB=$ROOT/$LIB
try() {  # name, method, unit, interval
  local name=$1 method=$2 unit=$3 interval=$4
  ( cd /tmp && timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit \
        --pc-sampling-interval $interval -d /tmp/rp_$name --output-format csv -- \
        python $ROOT/scripts/time_forward.py --n 50 --batch 256 --iters 300 ) > $OUT/pc_$name.log 2>&1
  echo "[$name] exit $?" >> $OUT/pc_$name.log
  find /tmp/rp_$name -type f 2>/dev/null | head -20 >> $OUT/pc_$name.log
  python $ROOT/scripts/pc_hist.py /tmp/rp_$name $OUT/pc_$name.csv >> $OUT/pc_$name.log 2>&1
  rm -rf /tmp/rp_$name
  tail -n 4 $OUT/pc_$name.log
}
try stoch_c1m stochastic cycles 1048576
try stoch_c64k stochastic cycles 65536
try host_t1 host_trap time 1
try host_t1000 host_trap time 1000
