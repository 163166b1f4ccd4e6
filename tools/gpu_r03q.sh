#!/bin/bash
mkdir -p gpurun_out/r03q
timeout 900 python tools/train_probe.py > gpurun_out/r03q/probe.txt 2>&1
tail -70 gpurun_out/r03q/probe.txt
