#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_train.py -m gpu -q -x 2>&1 | tail -5 > $O/tests.log; cat $O/tests.log
timeout 600 python bench.py --train --train-height 576 --train-width 1024 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/train_576x1024.json; cut -c1-260 $O/train_576x1024.json
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python bench.py --train --train-height 576 --train-width 1024 --steps 2 --warmup 1 > $O/ks.log 2>&1
python tools/rocprof_summary.py $(find /tmp/prof_tr -name "*results.db" | head -1) $O/r03o_train_step_kernel_stats.csv
head -24 $O/r03o_train_step_kernel_stats.csv
