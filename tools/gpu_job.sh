#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git); the committed form at round end is the full check: GPU suite (-rP), smoke, default bench line -> gpurun_out/TAG/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python tools/attn32_probe.py 2>&1 | tee $O/${TAG}_attn32_probe.txt
SPIKE=1 SHAPES=2304x10 timeout 300 python tools/attn32_probe.py 2>&1 | tee -a $O/${TAG}_attn32_probe.txt
