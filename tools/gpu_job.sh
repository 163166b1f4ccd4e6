#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git); the committed form is the round-end check: full GPU suite (-rP: the measured parity values of every passing test),
# smoke, default bench line -> gpurun_out/TAG/; `bash tools/profile_run.sh TAG` adds the rocprofv3 set.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -rP > $O/${TAG}_gpu_suite_full.log 2>&1
grep -n "passed\|failed\|rror" $O/${TAG}_gpu_suite_full.log | tail -6
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench"; T0=$SECONDS; timeout 900 python bench.py 2>/dev/null | tail -1 > $O/${TAG}_bench.json; cut -c1-260 $O/${TAG}_bench.json; echo "bench.py wall: $((SECONDS - T0)) s"
