#!/bin/bash
mkdir -p gpurun_out/r03g
timeout 600 python -m pytest tests/test_hip_train.py -q -s -x -k "unet_training_step" > gpurun_out/r03g/tests.log 2>&1; grep -E "f2\]|passed|failed|Error|error|assert|line [0-9]+" gpurun_out/r03g/tests.log | tail -30
