#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
echo "== fp16"; timeout 600 python bench.py --dtype fp16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_fp16.json; cut -c1-200 $O/${TAG}_bench_fp16.json
echo "== bf16"; timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_bf16.json; cut -c1-200 $O/${TAG}_bench_bf16.json
