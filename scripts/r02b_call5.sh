#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
for t in 1 4; do timeout 300 python scripts/phase_timeline.py --n 50 --batch 64 --team $t > $O/timeline_b64_team$t.log 2>&1; grep -E "^forward" $O/timeline_b64_team$t.log; grep -A17 "wave 0: 1" $O/timeline_b64_team$t.log; done
