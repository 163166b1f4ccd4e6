#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python tools/atiled_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/atiled_probe.log
