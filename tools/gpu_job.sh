#!/bin/bash
# per-call GPU job of the current experiment (edited per call; see git log for earlier forms)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "attn" 2>&1 | tail -2 | tee $O/tests.log
WIW_ATTN_8WAVES=2 timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_served_width.py -q -m gpu -k "attn or spatial" 2>&1 | tail -2 | tee -a $O/tests.log
for r in 1 2; do
echo "-- 4 waves, 64 keys / barrier (production)"; timeout 300 python tools/attn_probe.py 2>&1 | grep -i spatial | head -2 | tee -a $O/attn_4_1.log
echo "-- 8 waves, 128 keys / barrier"; WIW_ATTN_8WAVES=2 timeout 300 python tools/attn_probe.py 2>&1 | grep -i spatial | head -2 | tee -a $O/attn_8_2.log
echo "-- 8 waves, 64 keys / barrier"; WIW_ATTN_8WAVES=1 timeout 300 python tools/attn_probe.py 2>&1 | grep -i spatial | head -2 | tee -a $O/attn_8_1.log
done
