#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
echo "== ffn tests"; timeout 300 python -m pytest tests/test_hip_ffn.py -x -q 2>&1 | tail -3 | tee $O/ffn_tests.log
echo "== probe (nt loads)"; ROUNDS=2 timeout 120 python tools/ffn_probe.py 2>&1 | grep BEST | tee $O/ffn_probe.txt
echo "== plain loads, same box"; WIW_LIB=tools/ablate/libwiw_plainld.so ONLY_FUSED=1 ROUNDS=2 timeout 120 python tools/ffn_probe.py 2>&1 | grep BEST | head -1 | tee -a $O/ffn_probe.txt
echo "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/gpu_tests.log
echo "== bench"; timeout 300 python bench.py --dump-shapes $O/shapes.txt 2>/dev/null | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json
