#!/bin/bash
# round 3, call 2: per-atom phases v2 - parity, timeline, forward and chain timing
O=gpurun_out/r3c2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cabi.py tests/test_gpu_flags.py -x -q -k "not full_batch_properties and not pocket" > $O/pytest_v2.log 2>&1; echo "pytest v2 exit $?"; tail -n 5 $O/pytest_v2.log
DIFFLINKER_HIP_LIB=build/lib_prof_v2.so timeout 300 python scripts/phase_timeline.py --n 50 --batch 64 --team 1 > $O/tl_v2_b64.log 2>&1
DIFFLINKER_HIP_LIB=build/lib_prof_v2.so timeout 300 python scripts/phase_timeline.py --n 50 --batch 256 > $O/tl_v2_b256.log 2>&1
for b in 64 256; do
  timeout 200 python scripts/time_forward.py --batch $b --team 1 > $O/tf_v2_b$b.log 2>&1
  timeout 200 python scripts/time_forward.py --batch $b --team 1 --precision fp32 > $O/tf_v2_fp32_b$b.log 2>&1
done
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --noise philox > $O/bench_v2.log 2>&1
tail -n 1 $O/tf_*.log
grep -h -o '"value": [0-9.]*' $O/bench_*.log
