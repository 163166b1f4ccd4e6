#!/bin/bash
# kernel-trace stats of one fine-tuning step (576x1024x14) + the whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03j
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python bench.py --train --steps 1 --warmup 1 --train-height 576 --train-width 1024 > gpurun_out/r03j/tr.log 2>&1
python tools/rocprof_summary.py $(find /tmp/prof_tr -name "*results.db" | head -1) gpurun_out/r03j/r03j_train_step_kernel_stats.csv
head -25 gpurun_out/r03j/r03j_train_step_kernel_stats.csv
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r03j/tests_all.log 2>&1; grep -E "passed|failed|error" gpurun_out/r03j/tests_all.log | tail -2
