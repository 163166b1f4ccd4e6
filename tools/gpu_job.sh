#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git); the committed form at round end is the round-end check: full GPU suite, smoke, default bench line -> gpurun_out/TAG/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
# r08d: LayerNorm row moments of the fused temporal block on the matrix pipe against the v_dot2c build (tstat0) and the
# all-MFMA resident build (tres1); whole suite first (new: manipulation, config 0, fp32 stream)
timeout 600 python -m pytest tests/test_hip_temporal_block.py tests/test_hip_fp16.py tests/test_hip_served_width.py -q -m gpu -rP -k "temporal" > $O/${TAG}_temporal_tests.log 2>&1
grep -n "passed\|failed\|rror" $O/${TAG}_temporal_tests.log | tail -5; grep "^\[parity" $O/${TAG}_temporal_tests.log | cut -c1-200
if grep -q "failed\|rror\|Abort" $O/${TAG}_temporal_tests.log; then echo "temporal tests failed: stop"; exit 1; fi
for lib in tools/ablate/libwiw_tstat0.so world-in-world_amd/libwiwsvd.so tools/ablate/libwiw_tres1.so tools/ablate/libwiw_tstat0.so world-in-world_amd/libwiwsvd.so tools/ablate/libwiw_tres1.so; do
  echo "== $lib" >> $O/${TAG}_temporal_probe.txt
  WIW_LIB=$lib ONLY_FUSED=1 ITERS=20 timeout 300 python -u tools/temporal_probe.py >> $O/${TAG}_temporal_probe.txt 2>&1
done
cat $O/${TAG}_temporal_probe.txt
for v in tstat0 new; do
  lib=world-in-world_amd/libwiwsvd.so; [ $v = tstat0 ] && lib=tools/ablate/libwiw_tstat0.so
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_$v -o pt -- env WIW_LIB=$lib ONLY_FUSED=1 ITERS=5 python tools/temporal_probe.py > $O/pmc_$v.log 2>&1
  python tools/pmc_summary.py $O/${TAG}_pmc_temporal_probe_$v.csv $(find /tmp/pmc_$v -name "*results.db")
  grep temporal $O/${TAG}_pmc_temporal_probe_$v.csv
done
timeout 1500 python -m pytest tests -q -m gpu -rP > $O/${TAG}_gpu_suite_full.log 2>&1
grep -n "passed\|failed\|rror" $O/${TAG}_gpu_suite_full.log | tail -6
grep "config 0\|manipulation" $O/${TAG}_gpu_suite_full.log | cut -c1-300
echo "== bench new"; timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${TAG}_bench.json; cut -c1-200 $O/${TAG}_bench.json
echo "== bench tstat0";  WIW_LIB=tools/ablate/libwiw_tstat0.so timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${TAG}_bench_tstat0.json; cut -c1-200 $O/${TAG}_bench_tstat0.json
python - <<PY
import json
for f in ("$O/${TAG}_bench.json", "$O/${TAG}_bench_tstat0.json"):
    d=json.loads(open(f).read()); print(f, d["value"], d["other_kernels"]["temporal_block"], d["gemm_kernels"])
PY
