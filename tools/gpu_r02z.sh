#!/bin/bash
mkdir -p gpurun_out/r02z
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_fp16.py tests/test_hip_served_width.py -m gpu -q -x > gpurun_out/r02z/tests.log 2>&1; tail -2 gpurun_out/r02z/tests.log
S="64512,5120,640,0,1 16128,10240,1280,0,1 258048,2560,320,0,1"
for lib in base new base new; do
  echo "== $lib" >> gpurun_out/r02z/probe.txt
  if [ $lib = base ]; then L=$PWD/tools/ablate/libwiw_base.so; else L=$PWD/world-in-world_amd/libwiwsvd.so; fi
  WIW_LIB=$L TILED=1 timeout 200 python tools/gemm_probe.py $S >> gpurun_out/r02z/probe.txt 2>&1
done
grep -E "^==|TFLOP" gpurun_out/r02z/probe.txt
for lib in base new; do
  if [ $lib = base ]; then L=$PWD/tools/ablate/libwiw_base.so; else L=$PWD/world-in-world_amd/libwiwsvd.so; fi
  WIW_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'])"
done
timeout 400 python bench.py 2>gpurun_out/r02z/bench.err | tail -1 > gpurun_out/r02z/bench.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02z/bench.json").read())
print(d["value"], d["ms_per_step"], d.get("cpu_baseline"))
PY
