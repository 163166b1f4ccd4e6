#!/bin/bash
O=gpurun_out/r03v; mkdir -p $O
ONLY=ops TUNE=1 timeout 600 python tools/train_probe.py > $O/ops_tuned.txt 2>&1; grep -v amdgpu.ids $O/ops_tuned.txt | head -34
