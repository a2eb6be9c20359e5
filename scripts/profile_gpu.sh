#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats            -> per-kernel durations (must agree with bench.py's HIP-event time)
#   2. --pmc (own passes, no trace flags) -> MFMA busy / clocks, then FETCH_SIZE, then WRITE_SIZE
# Everything lands in gpurun_out/prof_$TAG/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r02}
shift || true
BENCH_ARGS=${*:---steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-traffic}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
run() {  # name, rocprof args...
  local name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 "$@" -d /tmp/rp_$name --output-format csv -- python $ROOT/bench.py $BENCH_ARGS ) > $OUT/$name.log 2>&1
  echo "[$name] exit $?" >> $OUT/$name.log
  # keep the small files only (gpurun merges at most 64 MiB back): the statistics, and of the per-dispatch tables the rows of
  # this library's kernels (the 1004 torch.randn launches of every chain make the raw tables hundreds of MB)
  for f in $(find /tmp/rp_$name -name '*.csv' 2>/dev/null); do
    case $f in
      *kernel_trace.csv|*counter_collection.csv) (head -1 $f; grep -E "sample_chain_fc|egnn_forward_fc|pk_|sampler_step|size_gnn" $f) > $OUT/${name}_$(basename $f) ;;
      *) cp $f $OUT/${name}_$(basename $f) ;;
    esac
  done
  rm -rf /tmp/rp_$name
}
run trace --kernel-trace --stats
# counter collection (rocprofv3 --pmc, ROCm 7.2) dies with a segmentation fault on hipLaunchCooperativeKernel - the second launch of a
# split chain, every team launch: the same kernel, grid and arguments through the plain launch API for these passes
export DIFFLINKER_TEAM_LAUNCH_PLAIN=1
run pmc_mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
ls -la $OUT
