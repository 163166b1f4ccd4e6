#!/bin/bash
# check of the round: whole GPU suite, smoke, default bench line, fine-tuning bench line + by-operator table
O=gpurun_out/r04n; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/tests_all.log 2>&1
grep -E "passed|failed|error" $O/tests_all.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py 2>/dev/null | tail -1 > $O/r04n_bench.json; cut -c1-200 $O/r04n_bench.json
timeout 600 python bench.py --train --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/r04n_train_step_576x1024.json; cut -c1-200 $O/r04n_train_step_576x1024.json
ONLY=ops TUNE=1 timeout 600 python tools/train_probe.py 2>&1 | grep -v amdgpu.ids > $O/r04n_train_step_by_operator.txt; grep "wall" $O/r04n_train_step_by_operator.txt
