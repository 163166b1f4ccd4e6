#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git); the committed form at round end is the round-end check: full GPU suite, smoke, default bench line -> gpurun_out/TAG/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
# r19f: fused FeedForward DMA schedules (FFN_W1_BULK / FFN_W2_SPLIT / FFN_BULK_AFTER variants), same-box probe, two rounds
for i in 1 2; do
for v in shipped bulk b2 b3 b4 b5; do
  if [ $v = shipped ]; then L=""; else L="WIW_LIB=tools/ablate/libwiw_$v.so"; fi
  echo "== $v: $(env $L ONLY_FUSED=1 ROUNDS=3 timeout 200 python tools/ffn_probe.py 2>&1 | grep BEST | tr '\n' ' ')"
done
done 2>&1 | tee $O/${TAG}_ffn_schedules.txt
for v in b2 b3 b4 b5; do echo "== tests $v: $(timeout 600 env WIW_LIB=tools/ablate/libwiw_$v.so python -m pytest tests/test_hip_ffn.py -q -m gpu -x -k 'not ffn32 and not fp16' 2>&1 | tail -1)"; done
