#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_team.py tests/test_gpu_parity.py tests/test_gpu_cabi.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python bench.py --batch 64 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120
