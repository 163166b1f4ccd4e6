#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_gemm16.py -x -q -s 2>&1 | grep -v "^$" | tail -12 | tee $O/tests.log
timeout 600 python tools/gemm16_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm16_probe.log
