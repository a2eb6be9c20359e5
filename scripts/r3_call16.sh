#!/bin/bash
mkdir -p gpurun_out/r3c16
timeout 900 python -m pytest tests/test_gpu_parity_hard.py tests/test_gpu_flags.py -x -q -m gpu -k "overflow or refused or sin" -s > gpurun_out/r3c16/pytest.log 2>&1
tail -15 gpurun_out/r3c16/pytest.log
