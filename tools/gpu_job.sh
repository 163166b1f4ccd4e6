#!/bin/bash
# per-call GPU job of the current experiment (edited per call; see git log for earlier forms)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1800 python -m pytest tests/test_hip_train.py -q -m gpu -x -s > $O/train_tests_full.log 2>&1
grep -n "\[f2\] forward log\|\[f2\] backward with\|passed\|failed\|rror" $O/train_tests_full.log | tail -14
echo "== train bench"; timeout 600 python bench.py --train --steps 3 --warmup 1 2>/dev/null | tail -1 | cut -c1-240 | tee $O/bench_train.log
