#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
echo "== tests"; timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py tests/test_hip_fp16.py tests/test_hip_served_width.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.log
for r in 1 2; do
echo "-- new (8 waves at S >= 2048)"; timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | grep -i spatial | tee -a $O/attn_new.log
echo "-- new, 4 waves"; WIW_ATTN_4WAVES=1 timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | grep -i spatial | head -2 | tee -a $O/attn_new4.log
echo "-- old"; WIW_LIB=tools/ablate/libwiw_attn_old.so timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | grep -i spatial | head -2 | tee -a $O/attn_old.log
done
