#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git); the committed form at round end is the round-end check: full GPU suite, smoke, default bench line -> gpurun_out/TAG/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
# r19i: the persistent K = 320 projection (thin.hip) — parity, probe, rollout A/B; FFN_FREE_K as the default; the reference's
# 25-step trajectory at the benchmarked size against the three precisions
timeout 900 python -m pytest tests/test_hip_thin.py -q -m gpu -x -s 2>&1 | tail -40 | tee $O/${TAG}_thin_tests.log
timeout 300 python tools/thin_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/${TAG}_thin_probe.txt
timeout 900 python -m pytest tests/test_hip_ffn.py -q -m gpu -x 2>&1 | tail -3 | tee $O/${TAG}_ffn_tests.log
timeout 1500 python -m pytest tests/test_hip_res32.py -q -m gpu -s -k "reference_trajectory_25_steps_at_the_benchmarked" 2>&1 | grep -v amdgpu.ids | tail -30 | tee $O/${TAG}_northstar_trajectory.log
for v in "" "WIW_NO_LINEAR320=1"; do
  echo "== ${v:-linear320}: $(env $v timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'],'frames/s', d['ms_per_step'],'ms | dense', r['achieved'],'TF/s frac',r['frac'],'| dense_split',d.get('dense_split'))")"
done 2>&1 | tee $O/${TAG}_rollout_ab.txt
