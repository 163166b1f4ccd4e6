#!/bin/bash
# round-end check of the final build: full GPU suite + smoke + profile set
mkdir -p gpurun_out/r02w
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r02w/tests_all.log 2>&1
tail -4 gpurun_out/r02w/tests_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02w/smoke.log 2>&1; tail -2 gpurun_out/r02w/smoke.log
bash tools/profile_run.sh r02w
