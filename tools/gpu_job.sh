#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git); the committed form at round end is the round-end check: full GPU suite, smoke, default bench line -> gpurun_out/TAG/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
# r19d: the general (Sk = 2..8) cross-attention: kernel vs torch, UNetHIP(num_past_obs=3) vs the reference fixture
timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_unet.py tests/test_hip_graph.py -q -m gpu -rP -x -k "cross_attention or conditioning or unet_tiny or transformer_golden or graph" > $O/${TAG}_tests.log 2>&1
grep -n "passed\|failed\|rror" $O/${TAG}_tests.log | tail -5; grep "\[parity\] cross\|3 conditioning" $O/${TAG}_tests.log | cut -c1-250
