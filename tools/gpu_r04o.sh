#!/bin/bash
O=gpurun_out/r04o; mkdir -p $O
for v in new old new old; do
  if [ $v = old ]; then export WIW_LIB=$PWD/tools/ablate/rules/libwiwsvd_oldrules.so; else unset WIW_LIB; fi
  echo "== $v rules: $(timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | cut -c1-100)"
done > $O/rules_ab.txt
cat $O/rules_ab.txt
