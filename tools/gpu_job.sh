#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git); the committed form at round end is the round-end check: full GPU suite, smoke, default bench line -> gpurun_out/TAG/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
# r08b: the fp32 residual stream (ABI 11): its tests with every measured value printed, then the whole suite (-rP: captured
# output of passing tests = the measured parity numbers), then the default bench line with `box` and `extra`
timeout 900 python -m pytest tests/test_hip_res32.py -q -m gpu -rP -k "not config0" > $O/${TAG}_res32_tests.log 2>&1
grep -n "passed\|failed\|rror" $O/${TAG}_res32_tests.log | tail -8
grep "tolerance\]\|res32 " $O/${TAG}_res32_tests.log | cut -c1-420
timeout 1500 python -m pytest tests -q -m gpu -rP --deselect tests/test_hip_res32.py > $O/${TAG}_gpu_suite_full.log 2>&1
grep -n "passed\|failed\|rror" $O/${TAG}_gpu_suite_full.log | tail -6
echo "== bench"; timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/${TAG}_bench.json; cut -c1-300 $O/${TAG}_bench.json; tail -3 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/${TAG}_bench.json").read())
print(json.dumps(d.get("box")), json.dumps(d.get("extra")), json.dumps(d.get("roofline"))[:400])
PY
