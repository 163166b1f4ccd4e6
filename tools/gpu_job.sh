#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
echo "== tests"; timeout 900 python -m pytest tests/test_hip_temporal_block.py -q -m gpu 2>&1 | tail -3 | tee $O/tests.log
WIW_TEMPORAL_RING=1 timeout 900 python -m pytest tests/test_hip_temporal_block.py -q -m gpu 2>&1 | tail -3 | tee -a $O/tests.log
for r in 1 2; do
echo "-- default"; ONLY_FUSED=1 timeout 300 python tools/temporal_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/temporal_default.log
echo "-- ring forced"; WIW_TEMPORAL_RING=1 SHAPES="2,14,9216,5" ONLY_FUSED=1 timeout 300 python tools/temporal_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/temporal_ring.log
done
echo "== wider tests"; timeout 1500 python -m pytest tests/test_hip_served_width.py tests/test_hip_fp16.py tests/test_hip_unet.py tests/test_hip_kernels.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O/tests.log
