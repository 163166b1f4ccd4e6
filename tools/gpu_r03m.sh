#!/bin/bash
mkdir -p gpurun_out/r03m
timeout 900 python -m pytest tests/test_hip_train.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r03m/tests.log
tail -5 gpurun_out/r03m/tests.log
timeout 600 python bench.py --train --steps 3 --warmup 1 > gpurun_out/r03m/train_256x512.json 2> gpurun_out/r03m/train_bench.err
timeout 600 python bench.py --train --train-height 576 --train-width 1024 --steps 3 --warmup 1 > gpurun_out/r03m/train_576x1024.json 2>> gpurun_out/r03m/train_bench.err
cat gpurun_out/r03m/train_256x512.json gpurun_out/r03m/train_576x1024.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03m/prof -o train -- python $GRAFT_REPO_ROOT/bench.py --train --train-height 576 --train-width 1024 --steps 2 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/r03m/prof/*/*kernel_stats.csv gpurun_out/r03m/prof/*kernel_stats.csv 2>/dev/null | head -1)
head -14 "$f"
find gpurun_out/r03m/prof -name "*.db" -delete; find gpurun_out/r03m/prof -name "*trace.csv" -delete
