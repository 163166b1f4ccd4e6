cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r10d; mkdir -p $O
echo "== halo test"; timeout 600 python -m pytest tests/test_hip_kernels.py -q -x -k "halo or channel_block_major" -rP 2>&1 | grep "halo\|passed\|failed\|rror\|assert" | head -30
echo "== probe"
for v in "KCMAJOR=1" "HALO=1" "KCMAJOR=1" "HALO=1"; do echo "-- $v"; env $v TILED=1 SERVED=1 timeout 200 python tools/gemm_probe.py 258048,320,2880,1 258048,320,5760,1 258048,320,8640,1 64512,640,5760,1 64512,640,11520,1 2>&1 | tail -5; done | tee $O/r10d_conv_halo_probe.txt
