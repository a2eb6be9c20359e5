#!/bin/bash
# round 6, final build: the round-5 differential fuzzers on new seeds (version-3 phases run for every f16x3 molecule of 33..55 atoms on one compute unit)
O=gpurun_out/r6/fuzz
mkdir -p $O
for s in 11 12 13; do timeout 900 python scripts/r5/fuzz_chain.py --check _fuzz/chain_s$s.pt > $O/chain_s$s.log 2>&1; tail -n 3 $O/chain_s$s.log; done
for s in 6 7; do timeout 1200 python scripts/r5/fuzz_forward.py --seed $s --cases 150 > $O/forward_s$s.log 2>&1; tail -n 4 $O/forward_s$s.log; done
