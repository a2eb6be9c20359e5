#!/bin/bash
# GPU box: replay one fuzz case under single-parameter overrides to see what its error follows.  usage: fuzz_locate.sh SEED CASE
s=$1; c=$2
for o in "" "precision=fp32" "attention=False" "hidden=128" "hidden=64" "trained=False" "mag=1.0" "sub=1" "L=1" "L=2" "coord_gain=0.02" "team=auto" "sizes=[28]" "sizes=[9]"; do
  echo "== override: ${o:-none}"
  timeout 120 python scripts/r5/fuzz_forward.py --seed $s --only $c ${o:+--set $o} 2>&1 | grep -E "^(ok|FAIL|skip)" | cut -c1-150
done
