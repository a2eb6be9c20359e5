#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_team.py -m gpu -q -s > $O/pytest_team.log 2>&1; echo "pytest(team) exit $?"; grep -E "^\[|passed|failed|Error|^FAILED" $O/pytest_team.log | cut -c1-220 | tail -40
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_team.py > $O/pytest_all.log 2>&1; echo "pytest(all, team auto) exit $?"; tail -12 $O/pytest_all.log | cut -c1-200
