#!/bin/bash
# A/B of HIP library builds on ONE box, several precisions: scripts/r4/ab_bench2.sh "<precisions>" lib1 lib2 ...  (twice, alternating)
precs=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    for p in $precs; do
      DIFFLINKER_HIP_LIB=build/lib_$lib.so timeout 900 python bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --precision $p 2>/dev/null | tail -n 1 | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', '$p', round(d['value'],1), 'mol/s  kernel_ms', round(d['roofline']['kernel_ms'],1))"
    done
  done
done
