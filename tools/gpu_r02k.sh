cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_hip_kernels.py tests/test_hip_temporal_block.py tests/test_hip_fullsize.py -m gpu -q --timeout 300 2>&1 | tail -3
export TILED=1
SH="258048,2560,320,0,1 258048,320,320 258048,640,320 64512,5120,640,0,1 64512,640,2560 258048,320,1280 16128,1280,5120 258048,320,2880,1 258048,320,8640,1 64512,640,5760,1 16128,1280,11520,1 258048,320,960,4 64512,640,1920,4"
echo "== burst (default)"; timeout 200 python tools/gemm_probe.py $SH 2>/dev/null | grep -v amdgpu
echo "== spread (round-1 schedule)"; WIW_LIB=$GRAFT_REPO_ROOT/tools/ablate/libwiw_spread.so timeout 200 python tools/gemm_probe.py $SH 2>/dev/null | grep -v amdgpu
echo "== temporal burst"; ONLY_FUSED=1 timeout 100 python tools/temporal_probe.py 2>/dev/null | grep -v amdgpu
echo "== temporal spread"; ONLY_FUSED=1 WIW_LIB=$GRAFT_REPO_ROOT/tools/ablate/libwiw_spread.so timeout 100 python tools/temporal_probe.py 2>/dev/null | grep -v amdgpu
timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120
WIW_LIB=$GRAFT_REPO_ROOT/tools/ablate/libwiw_spread.so timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120
