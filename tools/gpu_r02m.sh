#!/bin/bash
# fused q|k|v + transpose vs swapped V^T GEMM
set -x
mkdir -p gpurun_out/r02m
timeout 900 python -m pytest tests/test_hip_served_width.py tests/test_hip_unet.py tests/test_hip_kernels.py -m gpu -x -q > gpurun_out/r02m/tests.log 2>&1
tail -5 gpurun_out/r02m/tests.log
timeout 300 python bench.py --steps 2 --warmup 1 > gpurun_out/r02m/bench_new.json 2> gpurun_out/r02m/bench_new.err
WIW_SWAPPED_VT=1 timeout 300 python bench.py --steps 2 --warmup 1 > gpurun_out/r02m/bench_old.json 2> gpurun_out/r02m/bench_old.err
timeout 300 python bench.py --steps 2 --warmup 1 > gpurun_out/r02m/bench_new2.json 2>> gpurun_out/r02m/bench_new.err
python - <<'PY'
import json
for n in ("new","old","new2"):
    try:
        d=json.loads(open(f"gpurun_out/r02m/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], {k:v for k,v in d.get("other_kernels",{}).items() if k in("transpose",)})
    except Exception as e: print(n, "ERR", e)
PY
