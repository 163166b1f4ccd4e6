#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests/test_ema.py tests/test_hip_vae.py tests/test_abi.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -15 | tee $O/tests.log
