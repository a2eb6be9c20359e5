#!/bin/bash
O=gpurun_out/r6/per_atom_$1
mkdir -p $O
for cfg in "--n 50 --batch 64 --team 1"; do
  for v in prof ko_dma ko_mfma ko_dsr ko_all; do
    tag=$(echo $cfg | tr -d ' -')
    DIFFLINKER_HIP_LIB=difflinker_amd/variants/lib_$v.so timeout 300 python scripts/phase_timeline.py $cfg > $O/${v}_$tag.log 2>&1
    echo "== $v $cfg"; grep "^forward\|wave 0 per-atom phases" $O/${v}_$tag.log
  done
done
