#!/bin/bash
# round 6, call 2: first run of the version-3 per-atom phases
O=gpurun_out/r6/call2
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -n 3 $O/smoke.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -x -k "forward_vs_oracle or golden" > $O/pytest_fwd.log 2>&1; echo "pytest fwd exit $?"; tail -n 5 $O/pytest_fwd.log
grep -h "rel-L2" $O/pytest_fwd.log | head -40
for lib in "" difflinker_amd/variants/lib_r5base.so; do
  echo "== lib: ${lib:-product}"
  DIFFLINKER_HIP_LIB=$lib timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('headline', round(d['value'],1), 'kernel_ms', round(d['roofline']['kernel_ms'],1), d.get('split_chain'))
"
done
