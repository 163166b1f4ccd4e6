#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s; mkdir -p $O
ONLY=ops TUNE=0 timeout 600 python tools/train_probe.py > $O/ops_model.txt 2>&1; head -8 $O/ops_model.txt
ONLY=ops TUNE=1 timeout 600 python tools/train_probe.py > $O/ops_tuned.txt 2>&1; head -8 $O/ops_tuned.txt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python bench.py --train --steps 2 --warmup 2 > $O/ks.log 2>&1
python tools/rocprof_summary.py $(find /tmp/prof_tr -name "*results.db" | head -1) $O/r03s_train_step_kernel_stats_autotune.csv
head -16 $O/r03s_train_step_kernel_stats_autotune.csv
