#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git); the committed form at round end is the full check: GPU suite (-rP), smoke, default bench line -> gpurun_out/TAG/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_graph.py tests/test_hip_fp16.py tests/test_hip_unet.py tests/test_hip_server.py tests/test_manipulation.py tests/test_hip_parallel.py -q -m gpu -rP > $O/${TAG}_gpu_partial.log 2>&1
grep -n "passed\|failed\|rror" $O/${TAG}_gpu_partial.log | tail -8
echo "== bench"; timeout 900 python bench.py 2>/dev/null | tail -1 > $O/${TAG}_bench.json; python - <<'PY'
import json,sys
d=json.loads(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r12i/r12i_bench.json").read())
print(d["value"], d["roofline"]["frac"]); print(json.dumps(d.get("extra"))[:1500])
PY
