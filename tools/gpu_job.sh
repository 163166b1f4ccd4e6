#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git).  This form: the one-pass GroupNorm of the two inner levels (ABI 15) against the two-launch path.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py tests/test_hip_served_width.py tests/test_hip_graph.py -q -m gpu -x -rP > $O/${TAG}_tests.log 2>&1
grep -n "passed\|failed\|rror\|gn one pass" $O/${TAG}_tests.log | tail -14
echo "== bench one pass"; timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${TAG}_bench.json
echo "== bench two launches"; WIW_GN_TWO_KERNELS=1 timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${TAG}_bench_two.json
python - $TAG <<'PY'
import json,sys
t=sys.argv[1]
for n in ("bench","bench_two"):
    j=json.load(open(f"gpurun_out/{t}/{t}_{n}.json"))
    print(n, j["value"], "groupnorm", j["other_kernels"]["groupnorm"], "power", j.get("power"))
PY
