#!/bin/bash
mkdir -p gpurun_out/r02u
timeout 1200 python -m pytest tests/test_hip_fp16.py -m gpu -q -s > gpurun_out/r02u/fp16.log 2>&1
grep -E "parity fp16|passed|failed|Error|error" gpurun_out/r02u/fp16.log | head -60
