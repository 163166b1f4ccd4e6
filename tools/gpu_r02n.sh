#!/bin/bash
# split-K: kernel tests, served-width parity, fullsize property tests, bench A/B
set -x
mkdir -p gpurun_out/r02n
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "splitk" > gpurun_out/r02n/tests_k.log 2>&1
tail -4 gpurun_out/r02n/tests_k.log
timeout 1200 python -m pytest tests/test_hip_served_width.py tests/test_hip_unet.py tests/test_hip_fullsize.py -m gpu -x -q > gpurun_out/r02n/tests_u.log 2>&1
tail -4 gpurun_out/r02n/tests_u.log
timeout 300 python bench.py --steps 2 --warmup 1 > gpurun_out/r02n/bench_new.json 2> gpurun_out/r02n/bench_new.err
WIW_NO_SPLITK=1 timeout 300 python bench.py --steps 2 --warmup 1 > gpurun_out/r02n/bench_old.json 2> gpurun_out/r02n/bench_old.err
timeout 300 python bench.py --steps 1 --warmup 1 --dump-shapes > gpurun_out/r02n/shapes.json 2> gpurun_out/r02n/shapes.txt
python - <<'PY'
import json
for n in ("new","old"):
    try:
        d=json.loads(open(f"gpurun_out/r02n/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"])
    except Exception as e: print(n, "ERR", e)
PY
grep -E "M=4032 " gpurun_out/r02n/shapes.txt | head -20
