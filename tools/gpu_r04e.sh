#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_train.py -m gpu -q -x -s 2>&1 | grep -E "passed|failed|Error|error|assert|wgrad" | tail -12 > $O/tests.log; cat $O/tests.log
timeout 900 python tools/train_ab.py > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt | head -6; grep -c ", 2, \|(2, " $O/ab.txt
ONLY=ops TUNE=1 timeout 600 python tools/train_probe.py > $O/ops_tuned.txt 2>&1; grep "backward wall" -A5 $O/ops_tuned.txt
