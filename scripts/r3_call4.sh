#!/bin/bash
# round 3, call 4: new parity tests, bench with secondaries, 2 ranks on one GPU through gloo
O=gpurun_out/r3c4; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_hard.py tests/test_gpu_team.py tests/test_gpu_flags.py tests/test_gpu_cabi.py -m gpu -x -q -s > $O/pytest_new.log 2>&1; echo "pytest exit $?"; tail -n 4 $O/pytest_new.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "live_coordinate_head or large or splits" > $O/pytest_new2.log 2>&1; echo "pytest2 exit $?"; tail -n 3 $O/pytest_new2.log
timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench_full.log 2>&1; echo "bench exit $?"
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 1 --backend gloo --no-cpu-baseline --no-secondary > $O/bench_2ranks_gloo.log 2>&1; echo "gloo bench exit $?"; tail -n 2 $O/bench_2ranks_gloo.log | cut -c1-400
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r3c4/bench_full.log') if x.startswith('{')][-1]
d=json.loads(l)
print('headline', round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'frac', round(d['roofline']['frac'],4), 'alg', round(d['roofline']['algorithmic']['frac'],4))
for s in d.get('secondary',[]):
    print(' ', s['tag'], s.get('compute_units_per_molecule'), round(s['molecules_per_s'],1), round(s['kernel_ms'] or 0,1), round(s['roofline_frac'],3))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
