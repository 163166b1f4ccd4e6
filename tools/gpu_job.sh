#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
echo "== train tests (measured deviations)"; timeout 1500 python -m pytest tests/test_hip_train.py -x -q -s -k "full_width or 576x1024 or matches_reference_gradients or other_noise" 2>&1 | grep "\[f2\|passed\|failed\|Error\|assert" | cut -c1-400 | tee $O/train_tests.log
echo "== torchrun tests"; timeout 900 python -m pytest tests/test_hip_parallel.py -x -q 2>&1 | tail -3 | tee $O/parallel_tests.log
