#!/bin/bash
# round 3, call 3: teams v2 (atom-interleaved ownership, Q exchange, molecules up to 110 atoms) - the whole GPU suite + timings
O=gpurun_out/r3c3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest exit $?"; tail -n 6 $O/pytest_all.log
for b in 64 256; do
  timeout 200 python scripts/time_forward.py --batch $b --team 1 > $O/tf_b${b}_team1.log 2>&1
done
timeout 200 python scripts/time_forward.py --batch 64 --team 4 > $O/tf_b64_team4.log 2>&1
timeout 200 python scripts/time_forward.py --batch 32 --team 8 > $O/tf_b32_team8.log 2>&1
timeout 200 python scripts/time_forward.py --batch 128 --team 2 > $O/tf_b128_team2.log 2>&1
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --noise philox > $O/bench.log 2>&1
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --noise philox --batch 64 > $O/bench_b64.log 2>&1
tail -n 1 $O/tf_*.log
grep -h -o '"value": [0-9.]*' $O/bench*.log
