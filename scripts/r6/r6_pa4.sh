#!/bin/bash
O=gpurun_out/r6/per_atom_$1
mkdir -p $O
for cfg in "--n 50 --batch 64 --team 4" "--n 30 --batch 64 --team 1" "--n 50 --batch 128 --team 2"; do
  for v in prof r5base_prof; do
    tag=$(echo $cfg | tr -d ' -')
    DIFFLINKER_HIP_LIB=difflinker_amd/variants/lib_$v.so timeout 300 python scripts/phase_timeline.py $cfg > $O/${v}_$tag.log 2>&1
    echo "== $v $cfg"; grep "^forward\|wave 0 per-atom phases" $O/${v}_$tag.log; sed -n '/--- wave 0/,/wave 1:/p' $O/${v}_$tag.log | grep -v "stream phase\|^wave" 
  done
done
