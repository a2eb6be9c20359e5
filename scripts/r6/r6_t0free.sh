#!/bin/bash
O=gpurun_out/r6/t0free
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_team.py tests/test_gpu_cabi.py tests/test_gpu_round6.py -q -x 2>&1 | tail -n 1
run() {
  python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'mol/s %.1f' % d['value'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'], 'traffic GB/chain %.1f' % ((d['roofline']['traffic'] or 0)/1e9), d['roofline']['traffic_note'][150:260])"
}
for lib in "" difflinker_amd/variants/lib_r5base.so "" difflinker_amd/variants/lib_r5base.so; do DIFFLINKER_HIP_LIB=$lib run "lib ${lib:-product}"; done | tee $O/ab.log
for cfg in "--batch 256 --team 1 --n 35" "--batch 256 --team 1 --n 44" "--batch 256 --team 1"; do
  for lib in "" difflinker_amd/variants/lib_r5base.so; do
    DIFFLINKER_HIP_LIB=$lib timeout 300 python scripts/time_forward.py --raw --iters 50 $cfg 2>&1 | tail -1
  done
done | tee $O/forward.log
