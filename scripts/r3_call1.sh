#!/bin/bash
# round 3, call 1: where the per-atom phases spend their time (finer phase timeline) + early-prefetch variant A/B
O=gpurun_out/r3c1; mkdir -p $O
for v in prof_base prof_early; do
  DIFFLINKER_HIP_LIB=build/lib_$v.so timeout 300 python scripts/phase_timeline.py --n 50 --batch 256 > $O/tl_${v}_b256.log 2>&1
  DIFFLINKER_HIP_LIB=build/lib_$v.so timeout 300 python scripts/phase_timeline.py --n 50 --batch 64 --team 1 > $O/tl_${v}_b64.log 2>&1
done
for b in 64 256; do
  timeout 200 python scripts/time_forward.py --batch $b --team 1 > $O/tf_product_b$b.log 2>&1
  DIFFLINKER_HIP_LIB=build/lib_early.so timeout 200 python scripts/time_forward.py --batch $b --team 1 > $O/tf_early_b$b.log 2>&1
done
DIFFLINKER_HIP_LIB=build/lib_early.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_early.log 2>&1; echo "pytest early exit $?"
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --noise philox > $O/bench_product.log 2>&1
DIFFLINKER_HIP_LIB=build/lib_early.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --noise philox > $O/bench_early.log 2>&1
tail -n 1 $O/tf_*.log; tail -n 3 $O/pytest_early.log
grep -h -o '"value": [0-9.]*' $O/bench_*.log
