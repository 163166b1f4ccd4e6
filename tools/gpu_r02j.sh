cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_kernels.py tests/test_hip_fullsize.py -m gpu -q --timeout 300 -k "gemm or conv or geglu" 2>&1 | tail -3
export TILED=1
echo "== default heuristics"; timeout 200 python tools/gemm_probe.py 258048,2560,320,0,1 258048,320,320 258048,640,320 258048,960,320 2>/dev/null | grep -v amdgpu
echo "== forced big"; WIW_GEMM_TILE=big timeout 200 python tools/gemm_probe.py 258048,320,1280 258048,320,2880,1 64512,640,5760,1 258048,320,960,4 2>/dev/null | grep -v amdgpu
WIW_LIB=$GRAFT_REPO_ROOT/tools/ablate/libwiw_trace.so WIW_GEMM_TILE=big timeout 120 python tools/trace_probe.py 258048,320,1280 2>&1 | grep -E "mean|total"
