#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
DIAG=1 REPEATS=10 CHECK=0 ITERS=5 timeout 600 python tools/attn32_probe.py 2>&1 | grep -v "amdgpu.ids" | cut -c1-160 | tee -a $O/${TAG}_attn32_repeat.txt
DIAG=1 REPEATS=10 CHECK=0 ITERS=2 FRAMES=4 SHAPES=256x1,1024x2,2304x8 timeout 600 python tools/attn32_probe.py 2>&1 | grep -v "amdgpu.ids" | cut -c1-160 | tee -a $O/${TAG}_attn32_repeat.txt
SPIKE=1 REPEATS=10 ITERS=2 FRAMES=4 SHAPES=256x1,1024x2,2304x8 timeout 600 python tools/attn32_probe.py 2>&1 | grep -v "amdgpu.ids" | cut -c1-160 | tee -a $O/${TAG}_attn32_repeat.txt
