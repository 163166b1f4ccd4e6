#!/bin/bash
mkdir -p gpurun_out/r03h
timeout 900 python -m pytest tests/test_hip_train.py -q -s -k "sharded_adamw_over_rccl" > gpurun_out/r03h/tests.log 2>&1; grep -E "f2\]|passed|failed|Error|error|assert|line [0-9]+" gpurun_out/r03h/tests.log | tail -30
