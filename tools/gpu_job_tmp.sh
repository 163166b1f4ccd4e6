cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r10i; mkdir -p $O
echo "== halo test"; timeout 600 python -m pytest tests/test_hip_kernels.py -q -x -k "halo" -rP 2>&1 | grep "up +\|passed\|failed\|rror\|assert" | head -20
echo "== probe"
for rep in 1 2; do for v in "KCMAJOR=1" "HALO=1"; do echo "-- $v"; env $v TILED=1 SERVED=1 timeout 200 python tools/gemm_probe.py 258048,640,5760,3 64512,1280,11520,3 2>&1 | tail -2; done; done | tee $O/r10i_upconv_halo_probe.txt
echo "== unet tests"; timeout 900 python -m pytest tests -q -m gpu -x -k "hip_unet or served or determinis" 2>&1 | grep "passed\|failed\|rror" | tail -3
echo "== bench"; timeout 900 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
echo "== bench no halo"; WIW_CONV_NO_HALO=1 timeout 900 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
