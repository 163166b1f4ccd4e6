#!/bin/bash
O=gpurun_out/r04h; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_train.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $O/tests.log; cat $O/tests.log
timeout 900 python tools/train_ab.py > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt | head -5; grep -c "measured (3" $O/ab.txt; grep -c "measured (2" $O/ab.txt
