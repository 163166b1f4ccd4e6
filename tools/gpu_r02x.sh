#!/bin/bash
# two-slot main loop of the 256x160 tile (tools/ablate/libwiw_slots2.so) vs the four-slot one
mkdir -p gpurun_out/r02x
WIW_LIB=$PWD/tools/ablate/libwiw_slots2.so timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "gemm or conv or geglu or tiled or splitk" > gpurun_out/r02x/tests.log 2>&1; tail -2 gpurun_out/r02x/tests.log
S="258048,2560,320,0,1 258048,320,320,0 258048,960,320,0 258048,320,1280,0 64512,1920,640,0 258048,320,576,1 4032,1280,3840,4"
for lib in base slots2 base slots2; do
  echo "== $lib" >> gpurun_out/r02x/probe.txt
  if [ $lib = base ]; then L=$PWD/world-in-world_amd/libwiwsvd.so; else L=$PWD/tools/ablate/libwiw_slots2.so; fi
  WIW_LIB=$L WIW_GEMM_TILE=big TILED=1 timeout 200 python tools/gemm_probe.py $S >> gpurun_out/r02x/probe.txt 2>&1
done
grep -E "^==|TFLOP" gpurun_out/r02x/probe.txt
for lib in base slots2; do
  if [ $lib = base ]; then L=$PWD/world-in-world_amd/libwiwsvd.so; else L=$PWD/tools/ablate/libwiw_slots2.so; fi
  WIW_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'])"
done
