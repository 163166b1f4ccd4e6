#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
echo "== train, forced dist, direct"; WIW_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 timeout 300 python bench.py --gpus 1 --train --tiny --train-height 128 --train-width 256 --steps 2 --warmup 1 --no-autotune > $O/train_direct.log 2>&1; grep -v "^$" $O/train_direct.log | tail -25
