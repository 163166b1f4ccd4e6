#!/bin/bash
# exact-sweep GEGLU epilogue (this build) vs tools/ablate/libwiw_base.so; bench with the self-checking cpu_baseline leg
mkdir -p gpurun_out/r02y
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_fp16.py -m gpu -q -x -k "geglu or gemm" > gpurun_out/r02y/tests.log 2>&1; tail -2 gpurun_out/r02y/tests.log
S="258048,2560,320,0,1 64512,5120,640,0,1 4032,10240,1280,0,1 520,2560,320,0,1"
for lib in base new base new; do
  echo "== $lib" >> gpurun_out/r02y/probe.txt
  if [ $lib = base ]; then L=$PWD/tools/ablate/libwiw_base.so; else L=$PWD/world-in-world_amd/libwiwsvd.so; fi
  WIW_LIB=$L WIW_GEMM_TILE=big TILED=1 timeout 200 python tools/gemm_probe.py $S >> gpurun_out/r02y/probe.txt 2>&1
done
grep -E "^==|TFLOP" gpurun_out/r02y/probe.txt
timeout 400 python bench.py 2>gpurun_out/r02y/bench.err | tail -1 > gpurun_out/r02y/bench.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02y/bench.json").read())
print(d["value"], d["ms_per_step"], d.get("cpu_baseline"))
PY
tail -3 gpurun_out/r02y/bench.err
