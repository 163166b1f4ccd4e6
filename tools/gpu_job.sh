#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git).  This form: the one-launch GroupNorm statistics (ABI 12) — probe (correctness, rerun-identical, counters, rates;
# loads-in-flight variants), tile probes of the two short-K dense shapes, the GroupNorm / determinism tests, a bench line.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
echo "== norm probe (default build: 4 loads in flight, 5 waves/SIMD)"; RPB=64,256 timeout 300 python tools/norm_probe.py 2>&1 | grep -v "^layernorm" | tee $O/${TAG}_norm_probe.txt
for v in gnu8 gnu2; do echo "== norm probe ($v)"; WIW_LIB=tools/ablate/libwiw_$v.so timeout 300 python tools/norm_probe.py 2>&1 | grep "gn_stats\|groupnorm" | tee -a $O/${TAG}_norm_probe_$v.txt; done
echo "== tile probes"
for t in huge big; do echo "-- WIW_GEMM_TILE=$t RES=1"; RES=1 WIW_GEMM_HUGE_ANYK=1 WIW_GEMM_TILE=$t timeout 200 python tools/gemm_probe.py 64512,640,640 16128,1280,1280 258048,320,320 4032,1280,1280 2>&1 | tail -4; done | tee $O/${TAG}_tile_probe.txt
echo "== tests"; timeout 1200 python -m pytest tests -q -m gpu -x -k "groupnorm or determinis or res32 or served or hip_unet or train" > $O/${TAG}_tests.log 2>&1; grep -n "passed\|failed\|rror" $O/${TAG}_tests.log | tail -5
echo "== bench"; timeout 900 python bench.py 2>/dev/null | tail -1 > $O/${TAG}_bench.json; cut -c1-260 $O/${TAG}_bench.json
python -c "
import json; d=json.loads(open('$O/${TAG}_bench.json').read()); print(d['value'], d['roofline']['frac'], d['dense_split']); print({k:(v.get('seconds'),v.get('launches')) for k,v in d['other_kernels'].items()})"
