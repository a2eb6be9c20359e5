#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_d.log 2>&1
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_final_d.log 2>&1; tail -3 gpurun_out/pytest_final_d.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_d.log 2>&1; tail -1 gpurun_out/smoke_d.log
timeout 600 python bench.py > gpurun_out/bench_full_d.log 2>&1; tail -1 gpurun_out/bench_full_d.log | cut -c1-260
timeout 300 python bench.py --noise philox --no-cpu-baseline > gpurun_out/bench_full_philox_d.log 2>&1; tail -1 gpurun_out/bench_full_philox_d.log | cut -c1-200
bash scripts/profile_gpu.sh r01d > gpurun_out/profile_r01d.log 2>&1; tail -2 gpurun_out/profile_r01d.log
timeout 300 python scripts/phase_timeline.py --n 50 > gpurun_out/timeline_final.log 2>&1; grep -E "^forward|^wave [07]:" gpurun_out/timeline_final.log
