#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
# r20g: fused FeedForward with the up-projection accumulators starting at the bias (FFN_BIAS_INIT): tests, same-box probe A/B
timeout 900 python -m pytest tests/test_hip_ffn.py -q -m gpu -x 2>&1 | tail -3 | tee $O/${TAG}_ffn_tests.log
for i in 1 2; do
for v in shipped nobias; do
  if [ $v = shipped ]; then L=""; else L="WIW_LIB=tools/ablate/libwiw_$v.so"; fi
  echo "== $v: $(env $L ONLY_FUSED=1 ROUNDS=3 timeout 200 python tools/ffn_probe.py 2>&1 | grep BEST | tr '\n' ' ')"
done
done 2>&1 | tee $O/${TAG}_ffn_bias_init.txt
timeout 1500 python -m pytest tests/test_hip_res32.py tests/test_hip_served_width.py tests/test_hip_unet.py -q -m gpu -x 2>&1 | tail -3 | tee -a $O/${TAG}_ffn_tests.log
