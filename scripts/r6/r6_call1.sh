#!/bin/bash
# round 6, first GPU call: the new tests, the files whose bars changed, baseline timelines and the headline on this box
O=gpurun_out/r6/call1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_cabi.py -q -s -x > $O/pytest_new.log 2>&1; echo "pytest new exit $?"; tail -n 3 $O/pytest_new.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -x > $O/pytest_parity.log 2>&1; echo "pytest parity exit $?"; tail -n 2 $O/pytest_parity.log
DIFFLINKER_HIP_LIB=build/libdifflinker_hip_profile.so timeout 300 python scripts/phase_timeline.py --n 50 --batch 256 --team 1 > $O/phase_B256_n50.log 2>&1; echo "timeline exit $?"
DIFFLINKER_HIP_LIB=build/libdifflinker_hip_profile.so timeout 300 python scripts/phase_timeline.py --n 50 --batch 64 --team 4 > $O/phase_B64_team4.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline --no-traffic > $O/bench_head.log 2> $O/bench_head.err; echo "bench exit $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r6/call1/bench_head.log') if x.startswith('{')][-1]
d=json.loads(l)
print('headline', round(d['value'],1), 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'frac', round(d['roofline']['frac'],4), d.get('split_chain'))
PY
grep -h "rel-L2\|C2 B=256" $O/pytest_new.log | head -20
