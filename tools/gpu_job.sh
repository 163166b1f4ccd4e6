#!/bin/bash
# per-call GPU job of the current experiment (edited per call; see git log for earlier forms)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/${TAG}_gpu_suite_full.log 2>&1
grep -n "passed\|failed\|rror" $O/${TAG}_gpu_suite_full.log | tail -6
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
