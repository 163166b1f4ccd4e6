#!/bin/bash
# per-call GPU job of the current experiment (edited per call; see git log for earlier forms)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
for r in 1 2; do
echo "-- default (126 VGPRs, 4 waves/SIMD)"; timeout 300 python tools/attn_probe.py 2>&1 | grep -i spatial | head -2 | tee -a $O/attn_default.log
echo "-- min 5 waves/SIMD (96 VGPRs, 27 spilled)"; WIW_LIB=tools/ablate/libwiw_aw5.so timeout 300 python tools/attn_probe.py 2>&1 | grep -i spatial | head -2 | tee -a $O/attn_w5.log
done
