#!/bin/bash
# End-of-round validation on the GPU box: build check, GPU suite (verbose: every measured error), smoke, the default bench line
# (secondaries, CPU + eager baselines, in-run traffic), `python bench.py --gpus 2 --backend gloo` started WITHOUT a launcher, the
# rocprofv3 evidence (kernel trace + separate --pmc passes)
O=gpurun_out/r6/final
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build exit $?"
timeout 1800 python -m pytest tests -m gpu -q -s > $O/pytest_gpu_verbose.log 2>&1; echo "pytest exit $?"; tail -n 2 $O/pytest_gpu_verbose.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -n 1 $O/smoke.log
timeout 2400 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench exit $?"
timeout 600 python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 > $O/bench_2ranks_selfspawn.log 2>&1; echo "bench --gpus 2 (self-spawned, gloo) exit $?"; tail -n 1 $O/bench_2ranks_selfspawn.log | cut -c1-300
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r6/final/bench.log') if x.startswith('{')][-1]
d=json.loads(l)
print('headline', round(d['value'],1), d['dtype'], 'ms/step', round(d['ms_per_step'],1), 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'frac', round(d['roofline']['frac'],4), 'of dense f16', round(d['roofline'].get('frac_of_dense_f16_peak',0),4), d.get('value_by_noise_source'))
print('traffic', d['roofline']['traffic'], (d['roofline']['traffic_note'] or '')[:200])
print('companions', d['roofline'].get('companions'))
for s in d.get('secondary',[]):
    print(' ', s['tag'], s.get('compute_units_per_molecule'), round(s['molecules_per_s'],1), round(s['kernel_ms'] or 0,1), 'F_min frac', round(s['roofline_frac'],3), 'executed', round(s['executed_frac'],3))
print('cpu', {k: d['cpu_baseline'].get(k) for k in ('value','cores','kind')}, 'c4', (d['cpu_baseline'].get('c4_pockets') or {}).get('value'))
print('eager', d['eager_rocm_baseline'].get('value'), d['eager_rocm_baseline'].get('ms_per_forward'), 'size_gnn', d.get('size_gnn'))
PY
bash scripts/profile_gpu.sh r06 > $O/profile_gpu.log 2>&1; echo "profile exit $?"
mkdir -p $O/prof; cp gpurun_out/prof_r06/*stats*.csv gpurun_out/prof_r06/*.log gpurun_out/prof_r06/*kernel_trace.csv gpurun_out/prof_r06/*counter_collection.csv $O/prof/ 2>/dev/null
ls $O/prof | head -30
