#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git); the committed form at round end is the full check: GPU suite (-rP), smoke, default bench line -> gpurun_out/TAG/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
echo "# attn_spatial32_kernel ablation builds (-DWIW_A32_ABLATE=bits: 1 no softmax VALU, 2 no K/V DMA + no tile barrier, 4 no LDS reads, 8 Q.K MFMAs independent, 16 no P.V MFMAs); timing only" > $O/${TAG}_attn32_ablation.txt
for v in world-in-world_amd/libwiwsvd.so tools/ablate/libwiw_ab1.so tools/ablate/libwiw_ab2.so tools/ablate/libwiw_ab4.so tools/ablate/libwiw_ab8.so tools/ablate/libwiw_ab16.so tools/ablate/libwiw_ab3.so tools/ablate/libwiw_ab5.so tools/ablate/libwiw_ab7.so world-in-world_amd/libwiwsvd.so; do
  echo "== $v" | tee -a $O/${TAG}_attn32_ablation.txt
  WIW_LIB=$v SHAPES=9216x5,2304x10 CHECK=0 timeout 300 python tools/attn32_probe.py 2>&1 | grep "v2" | tee -a $O/${TAG}_attn32_ablation.txt
done
