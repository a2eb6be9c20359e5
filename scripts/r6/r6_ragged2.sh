#!/bin/bash
# the ragged slot plan once more, on top of the final build: the kernel's effect on the chain with the SAME hand-over plan, then with the plan's cost model told about it
O=gpurun_out/r6/ragged2
P=difflinker_amd/variants/lib_prev.so
R=difflinker_amd/variants/lib_ragged.so
mkdir -p $O
run() {
  python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'mol/s %.1f' % d['value'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'], d.get('split_chain'))"
}
for rep in 1 2 3; do
  DIFFLINKER_RAGGED_COST=0 DIFFLINKER_HIP_LIB=$P run "plain kernel, plain cost model  "
  DIFFLINKER_RAGGED_COST=0 DIFFLINKER_HIP_LIB=$R run "ragged kernel, plain cost model "
  DIFFLINKER_RAGGED_COST=1 DIFFLINKER_HIP_LIB=$R run "ragged kernel, ragged cost model"
done | tee $O/ab.log
