#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_vae.py tests/test_hip_unet.py -q -m gpu -x 2>&1 | tail -4 | tee $O/tests.log
for r in 1 2; do
  echo "-- two launches"; timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['other_kernels']['groupnorm'])" | tee -a $O/gn_two.log
  echo "-- three launches"; WIW_GN_THREE_LAUNCH=1 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['other_kernels']['groupnorm'])" | tee -a $O/gn_three.log
done
