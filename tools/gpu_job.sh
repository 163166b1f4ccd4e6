#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
echo "# ffn32_kernel (second schedule) ablation builds (-DFF_ABLATE=bits: 1 no GEGLU VALU, 2 no weight DMA in the loop, 4 no LDS fragment reads, 8 no barriers); timing only" > $O/${TAG}_ffn32_ablation.txt
for v in world-in-world_amd/libwiwsvd.so tools/ablate/libwiw_ffab1.so tools/ablate/libwiw_ffab2.so tools/ablate/libwiw_ffab4.so tools/ablate/libwiw_ffab8.so tools/ablate/libwiw_ffab3.so; do
  echo "== $v" | tee -a $O/${TAG}_ffn32_ablation.txt
  WIW_LIB=$v MS=258048 timeout 300 python tools/ffn32_probe.py 2>&1 | grep "ffn ffn32" | tee -a $O/${TAG}_ffn32_ablation.txt
done
