#!/bin/bash
mkdir -p gpurun_out/r03c
timeout 600 python -m pytest tests/test_hip_train.py tests/test_abi.py -q > gpurun_out/r03c/tests.log 2>&1; tail -5 gpurun_out/r03c/tests.log
