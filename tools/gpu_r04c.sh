#!/bin/bash
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_train.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $O/tests.log; cat $O/tests.log
timeout 600 python bench.py --train --steps 4 --warmup 2 2>/dev/null | tail -1 > $O/train_576x1024.json; cut -c1-200 $O/train_576x1024.json
ONLY=ops TUNE=1 timeout 600 python tools/train_probe.py > $O/ops_tuned.txt 2>&1; grep -v amdgpu.ids $O/ops_tuned.txt | sed -n 1,6p; grep "backward wall" -A5 $O/ops_tuned.txt
