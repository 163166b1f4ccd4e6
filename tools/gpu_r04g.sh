#!/bin/bash
O=gpurun_out/r04g; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_train.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $O/tests.log; cat $O/tests.log
timeout 600 python bench.py --train --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/train_576x1024.json; cut -c1-200 $O/train_576x1024.json
timeout 600 python bench.py --train --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/train_576x1024_b.json; cut -c1-200 $O/train_576x1024_b.json
