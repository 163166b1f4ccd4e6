#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
SHAPES="64512,5120,640,0,1 16128,10240,1280,0,1 258048,2560,320,0,1"
for r in 1 2; do
echo "== GEGLU epilogue new (packed) round $r"; TILED=1 timeout 300 python tools/gemm_probe.py $SHAPES 2>&1 | tail -4 | tee -a $O/gepk_new.log
echo "== GEGLU epilogue old round $r"; WIW_LIB=tools/ablate/libwiw_gepk0.so TILED=1 timeout 300 python tools/gemm_probe.py $SHAPES 2>&1 | tail -4 | tee -a $O/gepk_old.log
done
echo "== tests"; timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_ffn.py tests/test_hip_unet.py tests/test_hip_fp16.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.log
