#!/bin/bash
O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_train.py -m gpu -q -x -k "attention_backward or training_step or trainer" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > $O/tests.log; cat $O/tests.log
for v in B X Y B X Y; do echo "== variant $v"; WIW_LIB=$PWD/tools/ablate/occ/libwiwsvd_$v.so timeout 300 python tools/attn_bwd_probe.py 2>&1 | grep attn_backward; done > $O/occ.txt
cat $O/occ.txt
