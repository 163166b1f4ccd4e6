#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
echo "== ffn tests"; timeout 300 python -m pytest tests/test_hip_ffn.py -x -q 2>&1 | tail -3 | tee $O/ffn_tests.log
for v in ring3 ring4; do echo "variant $v"; WIW_LIB=tools/ablate/libwiw_$v.so ONLY_FUSED=1 ROUNDS=2 timeout 120 python tools/ffn_probe.py 2>&1 | grep BEST | head -1; done | tee $O/ffn_ring.txt
echo "base (ring 5)"; ROUNDS=2 timeout 120 python tools/ffn_probe.py 2>&1 | grep BEST | tee -a $O/ffn_ring.txt
for v in trace trace_r3; do echo "== $v"; WIW_LIB=tools/ablate/libwiw_$v.so timeout 120 python tools/ffn_trace.py > $O/ffn_$v.txt 2>&1; grep -v amdgpu.ids $O/ffn_$v.txt | head -19; done
