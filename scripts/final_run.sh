#!/bin/bash
# End-of-round validation on the GPU box: build check, GPU suite, smoke, the default bench line (with secondaries and CPU baseline)
mkdir -p gpurun_out
O=gpurun_out
python -c "import __graft_entry__ as g; g.build()" > $O/build_final.log 2>&1; echo "build exit $?"
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_final.log 2>&1; echo "pytest exit $?"; tail -2 $O/pytest_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_final.log 2>&1; echo "smoke exit $?"; tail -2 $O/smoke_final.log
timeout 1500 python bench.py > $O/bench_final.log 2>&1; echo "bench exit $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_final.log') if x.startswith('{')][-1]
d=json.loads(l)
print('headline', round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'frac', round(d['roofline']['frac'],4))
for s in d.get('secondary',[]):
    print(' ', s['tag'], s.get('compute_units_per_molecule'), round(s['molecules_per_s'],1), round(s['kernel_ms'] or 0,1), round(s['roofline_frac'],3))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
