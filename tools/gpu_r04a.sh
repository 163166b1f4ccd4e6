#!/bin/bash
O=gpurun_out/r04a; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_train.py -m gpu -q -x -s -k "other_noise or attention_backward" 2>&1 | grep -E "passed|failed|Error|error|assert|scenario" | tail -12 > $O/tests.log; cat $O/tests.log
