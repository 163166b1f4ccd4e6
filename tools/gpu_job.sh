#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git); the committed form at round end is the round-end check: full GPU suite, smoke, default bench line -> gpurun_out/TAG/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
# r08a: the spill-free 256x320 / 256x160 epilogues against the round-3 library on ONE box
SHAPES="258048,320,320 258048,960,320 64512,640,640 64512,1920,640 64512,640,2560 16128,1280,1280 16128,3840,1280 16128,1280,5120 64512,5120,640,0,1 16128,10240,1280,0,1 258048,320,2880,1 64512,640,5760,1 16128,1280,11520,1 64512,640,1920,4 16128,1280,3840,4 64512,640,5760,3"
for lib in tools/ablate/libwiw_r3.so world-in-world_amd/libwiwsvd.so; do
  echo "== $lib" >> $O/${TAG}_gemm_probe.txt
  WIW_LIB=$lib TILED=1 RES=1 ITERS=30 timeout 300 python tools/gemm_probe.py $SHAPES >> $O/${TAG}_gemm_probe.txt 2>&1
done
cat $O/${TAG}_gemm_probe.txt
timeout 1500 python -m pytest tests -q -m gpu -x > $O/${TAG}_gpu_suite_full.log 2>&1
grep -n "passed\|failed\|rror" $O/${TAG}_gpu_suite_full.log | tail -6
echo "== bench new"; timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench.json; cut -c1-400 $O/${TAG}_bench.json
echo "== bench r3";  WIW_LIB=tools/ablate/libwiw_r3.so timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_r3lib.json; cut -c1-200 $O/${TAG}_bench_r3lib.json
# WRITE_SIZE / FETCH_SIZE per kernel, 1 step, eager launches
i=0
for grp in "WRITE_SIZE" "FETCH_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc$i -o p$i -- python bench.py --num-inference-steps 1 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-graph > $O/pmc$i.log 2>&1
done
python tools/pmc_summary.py $O/${TAG}_pmc_1step.csv $(find /tmp/pmc1 /tmp/pmc2 -name "*results.db")
grep -i "gemm" $O/${TAG}_pmc_1step.csv | cut -c1-300
