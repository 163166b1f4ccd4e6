#!/bin/bash
O=gpurun_out/r04k; mkdir -p $O
for t in default big default big; do
  if [ $t = big ]; then export WIW_GEMM_TILE=big; else unset WIW_GEMM_TILE; fi
  echo "== train, tile rule $t: $(timeout 600 python bench.py --train --steps 4 --warmup 2 --no-autotune 2>/dev/null | tail -1 | cut -c100-200)"
done > $O/tile_ab.txt
for t in default big; do
  if [ $t = big ]; then export WIW_GEMM_TILE=big; else unset WIW_GEMM_TILE; fi
  echo "== inference, tile rule $t: $(timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | cut -c1-120)"
done >> $O/tile_ab.txt
cat $O/tile_ab.txt
