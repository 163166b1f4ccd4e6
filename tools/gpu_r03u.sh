#!/bin/bash
O=gpurun_out/r03u; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_train.py -m gpu -q -x -s -k "groupnorm or training_step or trainer" 2>&1 | grep -E "passed|failed|Error|error|assert|groupnorm" | tail -15 > $O/tests.log; cat $O/tests.log
ONLY=ops TUNE=1 timeout 600 python tools/train_probe.py > $O/ops_tuned.txt 2>&1; head -9 $O/ops_tuned.txt
timeout 600 python bench.py --train --steps 4 --warmup 2 2>/dev/null | tail -1 > $O/train_576x1024.json; cut -c1-200 $O/train_576x1024.json
