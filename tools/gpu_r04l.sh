#!/bin/bash
O=gpurun_out/r04l; mkdir -p $O
SH="8064,1280,1280 8064,1280,5120 8064,10240,1280 8064,3840,1280 8064,1280,3840,4 8064,1280,11520,1 2016,1280,1280 2016,1280,5120 2016,10240,1280 2016,1280,11520,1 32256,640,640 32256,640,2560 32256,5120,640"
for t in auto small big huge; do
  if [ $t = auto ]; then unset WIW_GEMM_TILE; else export WIW_GEMM_TILE=$t; fi
  echo "== WIW_GEMM_TILE=$t"; ITERS=10 timeout 300 python tools/gemm_probe.py $SH 2>&1 | grep -v amdgpu.ids
done > $O/tile_probe.txt
cat $O/tile_probe.txt
