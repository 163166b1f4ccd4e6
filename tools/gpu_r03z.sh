#!/bin/bash
O=gpurun_out/r03z; mkdir -p $O
for v in A B C A B C; do echo "== variant $v"; WIW_LIB=$PWD/tools/ablate/occ/libwiwsvd_$v.so timeout 300 python tools/attn_bwd_probe.py 2>&1 | grep attn_backward; done > $O/occ.txt
cat $O/occ.txt
