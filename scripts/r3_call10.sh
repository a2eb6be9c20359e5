#!/bin/bash
# round 3, call 10: sin_embedding on the HBM-resident kernels
mkdir -p gpurun_out/r3c10
timeout 900 python -m pytest tests/test_gpu_flags.py -x -q -m gpu -s > gpurun_out/r3c10/pytest_flags.log 2>&1
echo "flags exit $?" >> gpurun_out/r3c10/pytest_flags.log
tail -30 gpurun_out/r3c10/pytest_flags.log
timeout 900 python -m pytest tests/test_gpu_parity_hard.py tests/test_gpu_parity.py -k "pocket or large" -x -q -m gpu > gpurun_out/r3c10/pytest_pocket.log 2>&1
tail -3 gpurun_out/r3c10/pytest_pocket.log
timeout 600 python bench.py --config C4 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/r3c10/bench_c4.log 2>&1
tail -1 gpurun_out/r3c10/bench_c4.log | cut -c1-400
