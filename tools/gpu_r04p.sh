#!/bin/bash
mkdir -p gpurun_out/r04p
timeout 600 python -m pytest tests/test_hip_train.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -4 > gpurun_out/r04p/tests.log; cat gpurun_out/r04p/tests.log
