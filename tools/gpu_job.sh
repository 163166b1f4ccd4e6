#!/bin/bash
# per-call GPU job of the current experiment (edited per call; see git log for earlier forms)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_graph.py tests/test_hip_parallel.py tests/test_hip_server.py -q -m gpu -x > $O/tests_full.log 2>&1
grep -n "passed\|failed\|rror" $O/tests_full.log | tail -6
