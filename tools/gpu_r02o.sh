#!/bin/bash
set -x
mkdir -p gpurun_out/r02o
S="4032,1280,11520,1 4032,1280,23040,1 4032,1280,14080,1 4032,1280,3840,4 4032,1280,5120,0 4032,1280,1280,0"
for sk in 1 2 3 4; do
  echo "== SPLITK=$sk" >> gpurun_out/r02o/probe.txt
  SPLITK=$sk TILED=1 timeout 200 python tools/gemm_probe.py $S >> gpurun_out/r02o/probe.txt 2>&1
done
echo "== tiles: small vs big at K=320 (L0)" >> gpurun_out/r02o/probe.txt
for t in big small; do
  echo "-- WIW_GEMM_TILE=$t" >> gpurun_out/r02o/probe.txt
  WIW_GEMM_TILE=$t TILED=1 timeout 200 python tools/gemm_probe.py 258048,2560,320,0,1 258048,320,320,0 258048,960,320,0 258048,320,1280,0 >> gpurun_out/r02o/probe.txt 2>&1
done
cat gpurun_out/r02o/probe.txt
