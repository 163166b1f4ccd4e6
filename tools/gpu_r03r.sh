#!/bin/bash
O=gpurun_out/r03r; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_train.py -m gpu -q -x -s 2>&1 | grep -E "passed|failed|Error|error|assert|wgrad" | tail -15 > $O/tests.log; cat $O/tests.log
timeout 600 python bench.py --train --steps 3 --warmup 2 2>/dev/null | tail -1 > $O/train_576x1024_autotune.json; cut -c1-200 $O/train_576x1024_autotune.json
timeout 600 python bench.py --train --steps 3 --warmup 2 --no-autotune 2>/dev/null | tail -1 > $O/train_576x1024_model.json; cut -c1-200 $O/train_576x1024_model.json
ONLY=sweep timeout 600 python tools/train_probe.py > $O/sweep.txt 2>&1; tail -30 $O/sweep.txt
