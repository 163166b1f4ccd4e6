#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git); the committed form at round end is the full check: GPU suite (-rP), smoke, default bench line -> gpurun_out/TAG/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_fp16.py tests/test_hip_served_width.py tests/test_hip_res32.py tests/test_hip_unet.py tests/test_hip_graph.py -q -m gpu -rP > $O/${TAG}_gpu_partial.log 2>&1
grep -n "passed\|failed\|rror" $O/${TAG}_gpu_partial.log | tail -8
grep "BENCHMARKED" $O/${TAG}_gpu_partial.log | cut -c1-420
echo "== bench"; timeout 900 python bench.py --no-extras 2>/dev/null | tail -1 > $O/${TAG}_bench.json; cut -c1-300 $O/${TAG}_bench.json
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU -d /tmp/pmc1 -o p1 -- env SHAPES=9216x5,2304x10 CHECK=0 ITERS=3 python tools/attn32_probe.py > $O/pmc1.log 2>&1
python tools/pmc_summary.py $O/${TAG}_pmc_attn.csv $(find /tmp/pmc1 -name "*results.db") ; grep -i "attn\|kernel," $O/${TAG}_pmc_attn.csv
