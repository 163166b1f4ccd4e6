#!/bin/bash
mkdir -p gpurun_out/r03k
timeout 900 python -m pytest tests/test_hip_train.py -q -x > gpurun_out/r03k/tests.log 2>&1; grep -E "passed|failed|rror" gpurun_out/r03k/tests.log | tail -2
timeout 600 python bench.py --train --steps 2 --warmup 1 --train-height 576 --train-width 1024 2>/dev/null | tail -1 > gpurun_out/r03k/r03k_train_576x1024.json; cut -c1-260 gpurun_out/r03k/r03k_train_576x1024.json
timeout 600 python bench.py --train --steps 2 --warmup 1 2>/dev/null | tail -1 > gpurun_out/r03k/r03k_train_256x512.json; cut -c1-260 gpurun_out/r03k/r03k_train_256x512.json
