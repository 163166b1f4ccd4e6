#!/bin/bash
mkdir -p gpurun_out/r03d
timeout 600 python -m pytest tests/test_hip_train.py -q -s > gpurun_out/r03d/tests.log 2>&1; grep -E "f2\]|passed|failed|rror|assert" gpurun_out/r03d/tests.log | tail -20
