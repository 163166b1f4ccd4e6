#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
# r20e: residual_fp32 = True (block level, the mode) / "full": every test that runs the modes + the bench legs
timeout 2400 python -m pytest tests -q -m gpu -rP -k "res32 or fp16 or graph or unet or served or server or fullsize" > $O/${TAG}_res32_tests.log 2>&1
grep -n "passed\|failed\|rror" $O/${TAG}_res32_tests.log | tail -8
grep -n "tolerance" $O/${TAG}_res32_tests.log | grep -i "stream" | cut -c1-330 | tail -30
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/${TAG}_bench.json; python -c "
import json; d=json.load(open('$O/${TAG}_bench.json')); print(d['value'], d['extra'])" | cut -c1-1500
