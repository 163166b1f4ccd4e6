#!/bin/bash
mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "layernorm_fold" > gpurun_out/r03a/tests_k.log 2>&1; grep -E "parity|passed|failed|rror" gpurun_out/r03a/tests_k.log | tail -12
timeout 1500 python -m pytest tests/test_hip_unet.py tests/test_hip_served_width.py tests/test_hip_fp16.py tests/test_hip_fullsize.py tests/test_hip_kernels.py -m gpu -q -x > gpurun_out/r03a/tests_u.log 2>&1; tail -3 gpurun_out/r03a/tests_u.log
for v in fold nofold fold nofold; do
  if [ $v = fold ]; then E=""; else E="WIW_LN_UNFOLDED=1"; fi
  env $E timeout 300 python bench.py --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done
