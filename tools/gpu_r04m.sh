#!/bin/bash
O=gpurun_out/r04m; mkdir -p $O
SH="8064,1280,1280 8064,1280,5120 8064,10240,1280 8064,3840,1280 8064,1280,3840,4 8064,1280,11520,1 2016,1280,1280 32256,640,640 32256,5120,640"
ITERS=10 timeout 300 python tools/gemm_probe.py $SH 2>&1 | grep -v amdgpu.ids > $O/tile_probe_new_rules.txt; cat $O/tile_probe_new_rules.txt
timeout 600 python bench.py --train --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/train.json; cut -c100-200 $O/train.json
timeout 600 python bench.py --train --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/train_b.json; cut -c100-200 $O/train_b.json
