#!/bin/bash
# round 6, final build: second fuzz campaign (chains: seeds 14-19, forwards: seeds 8-12)
O=gpurun_out/r6/fuzz2
mkdir -p $O
for s in 14 15 16 17 18 19; do timeout 900 python scripts/r5/fuzz_chain.py --check _fuzz/chain_s$s.pt > $O/chain_s$s.log 2>&1; tail -n 1 $O/chain_s$s.log; done
for s in 8 9 10 11 12; do timeout 1200 python scripts/r5/fuzz_forward.py --seed $s --cases 150 > $O/forward_s$s.log 2>&1; grep -a "cases in\|^FAIL" $O/forward_s$s.log | cut -c1-300; done
