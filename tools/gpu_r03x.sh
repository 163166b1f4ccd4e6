#!/bin/bash
# check of the round: whole GPU suite, smoke, default bench line
mkdir -p gpurun_out/r03x
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r03x/tests_all.log 2>&1
grep -E "passed|failed|error" gpurun_out/r03x/tests_all.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03x/smoke.log 2>&1; tail -1 gpurun_out/r03x/smoke.log
timeout 400 python bench.py 2>/dev/null | tail -1 > gpurun_out/r03x/r03x_bench.json; cut -c1-260 gpurun_out/r03x/r03x_bench.json
