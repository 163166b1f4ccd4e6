#!/bin/bash
O=gpurun_out/r03y; mkdir -p $O
timeout 600 python bench.py --train --train-height 256 --train-width 512 --steps 4 --warmup 2 2>/dev/null | tail -1 > $O/r03y_train_step_256x512.json; cut -c1-200 $O/r03y_train_step_256x512.json
timeout 600 python bench.py --train --dtype fp16 --steps 4 --warmup 2 2>/dev/null | tail -1 > $O/r03y_train_step_576x1024_fp16.json; cut -c1-200 $O/r03y_train_step_576x1024_fp16.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python bench.py --train --steps 2 --warmup 2 --no-autotune > $O/ks.log 2>&1
python tools/rocprof_summary.py $(find /tmp/prof_tr -name "*results.db" | head -1) $O/r03y_train_step_kernel_stats.csv
head -20 $O/r03y_train_step_kernel_stats.csv
