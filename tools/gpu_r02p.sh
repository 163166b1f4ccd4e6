#!/bin/bash
set -x
mkdir -p gpurun_out/r02p
timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "splitk" > gpurun_out/r02p/tests_k.log 2>&1
tail -3 gpurun_out/r02p/tests_k.log
S="4032,1280,11520,1 4032,1280,23040,1 4032,1280,3840,4 4032,1280,5120,0"
for sk in 1 2 3 4 5 6; do
  echo "== SPLITK=$sk (auto tile)" >> gpurun_out/r02p/probe.txt
  SPLITK=$sk TILED=1 timeout 200 python tools/gemm_probe.py $S >> gpurun_out/r02p/probe.txt 2>&1
done
grep -E "^==|TFLOP|rror" gpurun_out/r02p/probe.txt
timeout 1200 python -m pytest tests/test_hip_served_width.py tests/test_hip_unet.py tests/test_hip_fullsize.py -m gpu -x -q > gpurun_out/r02p/tests_u.log 2>&1
tail -3 gpurun_out/r02p/tests_u.log
timeout 300 python bench.py --steps 2 --warmup 1 > gpurun_out/r02p/bench_new.json 2> gpurun_out/r02p/bench_new.err
WIW_NO_SPLITK=1 timeout 300 python bench.py --steps 2 --warmup 1 > gpurun_out/r02p/bench_old.json 2> gpurun_out/r02p/bench_old.err
python - <<'PY'
import json
for n in ("new","old"):
    try:
        d=json.loads(open(f"gpurun_out/r02p/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"])
    except Exception as e: print(n, "ERR", e)
PY
