#!/bin/bash
O=gpurun_out/r3c9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_flags.py tests/test_gpu_philox.py -x -q -s > $O/pytest_flags.log 2>&1; echo "exit $?"; tail -n 5 $O/pytest_flags.log; grep "pocket flags\|120-atom" $O/pytest_flags.log | head -12
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "pocket or large or splits" > $O/pytest_pk.log 2>&1; echo "exit $?"; tail -n 2 $O/pytest_pk.log
