#!/bin/bash
# round 3, call 12: team exchange in one round trip + chunked slot-partial sums: team tests, then B = 64 / C2L / C2 timings
mkdir -p gpurun_out/r3c12
timeout 1200 python -m pytest tests/test_gpu_team.py tests/test_gpu_parity.py tests/test_gpu_flags.py -x -q -m gpu > gpurun_out/r3c12/pytest.log 2>&1
tail -3 gpurun_out/r3c12/pytest.log
for a in "--batch 64" "--config C2L" ""; do
  timeout 600 python bench.py $a --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$a', round(d['value'],1), round(d['ms_per_step'],1))" | tee -a gpurun_out/r3c12/bench.log
done
