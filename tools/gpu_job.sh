#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git); the committed form at round end is the round-end check: full GPU suite, smoke, default bench line -> gpurun_out/TAG/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
# r09c: width of the per-XCD super-tile of the 256x320 kernel (WIW_GEMM_SN) per shape
SHAPES="64512,5120,640,0,1 16128,10240,1280,0,1 64512,640,2560 16128,1280,5120 64512,1920,640 16128,3840,1280 64512,640,640 16128,1280,1280 258048,320,2880,1 64512,640,5760,1 16128,1280,11520,1 64512,640,1920,4 16128,1280,3840,4"
for sn in 0 8 4 2 1 0 4; do
  echo "== WIW_GEMM_SN=$sn" >> $O/${TAG}_sn_sweep.txt
  e=""; [ $sn != 0 ] && e="WIW_GEMM_SN=$sn"
  env $e KCMAJOR=1 TILED=1 RES=1 ITERS=30 timeout 300 python -u tools/gemm_probe.py $SHAPES >> $O/${TAG}_sn_sweep.txt 2>&1
done
grep -v amdgpu $O/${TAG}_sn_sweep.txt
