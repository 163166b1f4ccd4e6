#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python bench.py --train --train-height 576 --train-width 1024 --steps 2 --warmup 1 > $O/ks.log 2>&1
python tools/rocprof_summary.py $(find /tmp/prof_tr -name "*results.db" | head -1) $O/r03n_train_step_kernel_stats.csv
head -30 $O/r03n_train_step_kernel_stats.csv
