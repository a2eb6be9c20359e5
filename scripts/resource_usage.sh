#!/bin/bash
# Per-kernel register / spill / scratch figures of the HIP sources (compile only, no GPU needed):
#   scripts/resource_usage.sh [extra hipcc flags, e.g. -DDL_SPLIT_TRUNC]
cd "$(dirname "$0")/.."
for f in egnn_fc egnn_sparse size_gnn; do
  extra=""; [ $f = egnn_fc ] && extra="-mllvm -amdgpu-sched-strategy=iterative-ilp"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I include $extra "$@" \
      -c difflinker_amd/csrc/$f.hip -o /tmp/ru_$f.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//' |
  awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /AGPRs:/ {a=$NF} /VGPRs Spill:/ {vs=$NF} /SGPRs Spill/ {ss=$NF} /ScratchSize/ {sc=$NF}
       /Occupancy/ {occ=$NF} /LDS Size/ {printf "%s vgpr %s agpr %s vspill %s sspill %s scratch %s occ %s lds %s\n", name, v, a, vs, ss, sc, occ, $NF}' |
  c++filt | sed 's/(anonymous namespace):://g; s/((anonymous namespace)::[A-Za-z]*)//g'
done
