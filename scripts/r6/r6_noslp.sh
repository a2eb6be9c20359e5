#!/bin/bash
# A/B: the product library against the same sources built with -fno-slp-vectorize (no v_pk_*_f32 beside the MFMAs of the pair loop)
O=gpurun_out/r6/noslp
V=${1:-difflinker_amd/variants/lib_noslp.so}
mkdir -p $O
DIFFLINKER_HIP_LIB=$V timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round6.py -q -x 2>&1 | tail -n 1
run() {
  python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'mol/s %.1f' % d['value'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'])"
}
for lib in "" $V "" $V "" $V; do DIFFLINKER_HIP_LIB=$lib run "lib ${lib:-product}"; done | tee $O/ab.log
for cfg in "--batch 256 --team 1 --n 35" "--batch 256 --team 1" "--batch 64 --team 4" "--batch 128 --team 2"; do
  for lib in "" $V; do
    DIFFLINKER_HIP_LIB=$lib timeout 300 python scripts/time_forward.py --raw --iters 50 $cfg 2>&1 | tail -1
  done
done | tee $O/forward.log
