cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export WIW_GEMM_TILE=big TILED=1
SH="258048,320,1280 258048,640,320 258048,320,2880,1"
echo "== base"; timeout 100 python tools/gemm_probe.py $SH 2>/dev/null | grep -v amdgpu
for n in 1 2 4 6 10; do echo "== ablate $n"; WIW_LIB=$GRAFT_REPO_ROOT/tools/ablate/libwiw_a$n.so timeout 100 python tools/gemm_probe.py $SH 2>/dev/null | grep -v amdgpu; done
