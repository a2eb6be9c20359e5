#!/bin/bash
# round 3, call 22: time + board power of Dynamics.forward (uniform n = 50) for the product and three knock-outs of the pair loop
# (KW: no W2' fragment reads from LDS; KM: second-layer MFMAs replaced by one VALU op each; KT: transcendentals replaced by plain ops)
mkdir -p gpurun_out/r3c22
for b in 64 256; do
  for lib in prod ko_KW ko_KM ko_KT prod; do
    DIFFLINKER_HIP_LIB=build/lib_$lib.so timeout 300 python scripts/power_forward.py --batch $b 2>/dev/null | tail -1 | sed "s/^/$lib /" | tee -a gpurun_out/r3c22/energy.log
  done
done
