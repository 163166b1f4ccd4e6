#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 300 python tools/norm_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/norm_probe.log
