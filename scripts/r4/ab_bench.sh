#!/bin/bash
# A/B of HIP library builds on ONE box: scripts/r4/ab_bench.sh "<bench args>" lib1 lib2 ...   (each build/lib_<name>.so, twice, alternating)
args=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    DIFFLINKER_HIP_LIB=build/lib_$lib.so timeout 900 python bench.py $args --no-secondary --no-cpu-baseline 2>/dev/null | tail -n 1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', round(d['value'],1), 'mol/s  kernel_ms', round(d['roofline']['kernel_ms'],1))"
  done
done
