#!/bin/bash
# round 3, call 21: attention as a kernel variant (the default kernels carry no spill-heavy attention loop): tests + timings
mkdir -p gpurun_out/r3c21
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3c21/pytest.log 2>&1; tail -2 gpurun_out/r3c21/pytest.log
for a in "" "--batch 64" "--config C2L"; do
  timeout 600 python bench.py $a --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$a', round(d['value'],1), round(d['ms_per_step'],1))" | tee -a gpurun_out/r3c21/bench.log
done
DIFFLINKER_HIP_LIB=build/lib_prod.so timeout 600 python bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('previous build', round(d['value'],1), round(d['ms_per_step'],1))" | tee -a gpurun_out/r3c21/bench.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('this build again', round(d['value'],1), round(d['ms_per_step'],1))" | tee -a gpurun_out/r3c21/bench.log
