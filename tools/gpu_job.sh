#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_temporal_block.py -q -m gpu 2>&1 | tail -4 | tee $O/tests.log
