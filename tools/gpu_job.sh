#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/${TAG}_gpu_suite_full.log 2>&1
grep -n "passed\|failed\|error\|Error" $O/${TAG}_gpu_suite_full.log | tail -15
