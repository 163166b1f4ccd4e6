#!/bin/bash
O=gpurun_out/r04d; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_train.py -m gpu -q -x -k "row_major_operands" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $O/tests.log; cat $O/tests.log
ONLY=sweep timeout 900 python tools/train_probe.py > $O/sweep.txt 2>&1; grep -v amdgpu.ids $O/sweep.txt | tail -60
