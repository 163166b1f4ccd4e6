#!/bin/bash
# round-end check: full GPU suite + smoke + profile set
set -x
mkdir -p gpurun_out/r02q
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02q/tests_all.log 2>&1
tail -6 gpurun_out/r02q/tests_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02q/smoke.log 2>&1; tail -2 gpurun_out/r02q/smoke.log
bash tools/profile_run.sh r02q
