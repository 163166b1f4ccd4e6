#!/bin/bash
# One gpurun call = one run of this script (edited per call; results under gpurun_out/$TAG/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
echo "== ffn tests"; timeout 300 python -m pytest tests/test_hip_ffn.py -x -q -s 2>&1 | tail -40 > $O/ffn_tests.log; tail -15 $O/ffn_tests.log
if grep -q "failed\|error\|Timeout\|Killed" $O/ffn_tests.log; then echo "ffn tests not green: stopping"; exit 1; fi
echo "== ffn probe"; timeout 200 python tools/ffn_probe.py > $O/ffn_probe.txt 2>&1; grep BEST $O/ffn_probe.txt
echo "== integration"; timeout 600 python -m pytest tests/test_hip_served_width.py tests/test_hip_unet.py -x -q 2>&1 | tail -5 | tee $O/integration.log
echo "== bench fused"; timeout 300 python bench.py --no-cpu-baseline --dump-shapes $O/shapes_fused.txt 2>/dev/null | tail -1 > $O/bench_fused.json; cut -c1-400 $O/bench_fused.json
echo "== bench unfused"; WIW_FF_UNFUSED=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_unfused.json; cut -c1-200 $O/bench_unfused.json
echo "== bench fused, LN pass kept"; WIW_FFN_NO_LN=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_fused_noln.json; cut -c1-200 $O/bench_fused_noln.json
