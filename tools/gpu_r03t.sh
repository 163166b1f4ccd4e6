#!/bin/bash
O=gpurun_out/r03t; mkdir -p $O
timeout 900 python tools/train_ab.py > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt | head -80
