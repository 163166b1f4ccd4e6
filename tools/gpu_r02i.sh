cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WIW_LIB=$GRAFT_REPO_ROOT/tools/ablate/libwiw_trace.so WIW_GEMM_TILE=big timeout 120 python tools/trace_probe.py 258048,320,1280 258048,640,320 2>&1 | grep -v amdgpu
