#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
for v in default aabl1 aabl2 aabl3 aabl12 aabl13 aabl15; do
  if [ $v = default ]; then unset WIW_LIB; else export WIW_LIB=tools/ablate/libwiw_$v.so; fi
  echo "-- $v"; timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids | grep -i spatial | head -2 | tee -a $O/attn_ablate.log
done
