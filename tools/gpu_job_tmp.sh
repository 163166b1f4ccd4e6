cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r10g; mkdir -p $O
echo "== halo test"; timeout 600 python -m pytest tests/test_hip_kernels.py -q -x -k "halo" -rP 2>&1 | grep "halo\|passed\|failed\|rror\|assert" | head -40
echo "== probe"
for rep in 1 2; do for v in "KCMAJOR=1" "HALO=1"; do echo "-- $v"; env $v TILED=1 SERVED=1 timeout 200 python tools/gemm_probe.py 16128,1280,11520,1 16128,1280,23040,1 2>&1 | tail -2;
env $v SC=640 TILED=1 SERVED=1 timeout 200 python tools/gemm_probe.py 258048,320,3520,1 2>&1 | tail -1; env $v SC=1280 TILED=1 SERVED=1 timeout 200 python tools/gemm_probe.py 64512,640,7040,1 2>&1 | tail -1; env $v SC=2560 TILED=1 SERVED=1 timeout 200 python tools/gemm_probe.py 16128,1280,14080,1 2>&1 | tail -1; done; done | tee $O/r10g_conv_halo_l2_sc_probe.txt
