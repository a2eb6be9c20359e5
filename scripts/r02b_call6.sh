#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python bench.py --no-cpu-baseline --steps 3 > $O/bench_full.log 2>&1; echo "bench exit $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_full.log') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for s in d.get('secondary',[]):
    print(s['tag'], s.get('compute_units_per_molecule'), round(s['molecules_per_s'],1), round(s['kernel_ms'] or 0,1), round(s['roofline_frac'],3))
PY
