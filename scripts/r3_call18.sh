#!/bin/bash
# round 3, call 18: rocprofv3 evidence (kernel trace + PMC passes) of the default bench command on the final build, then the
# secondary workloads' kernel statistics
bash scripts/profile_gpu.sh r03 > gpurun_out/profile_r03.log 2>&1
tail -5 gpurun_out/profile_r03.log
bash scripts/profile_secondary.sh > /dev/null 2>&1
ls gpurun_out/prof_secondary | head
