#!/bin/bash
mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests/test_hip_unet.py tests/test_hip_served_width.py -m gpu -q -s -k "folded or full_width_golden or tiny_b1" > gpurun_out/r03b/tests.log 2>&1; grep -E "parity|passed|failed|rror" gpurun_out/r03b/tests.log | tail -12
timeout 300 python bench.py --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'])"
