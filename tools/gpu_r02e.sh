cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_kernels.py tests/test_hip_served_width.py -m gpu -q -k "attn_spatial" --timeout 300 2>&1 | tail -2
echo "== pipelined"; timeout 120 python tools/attn_probe.py 2>/dev/null | grep -v amdgpu
echo "== round-1 kernel"; WIW_ATTN_PIPE=0 timeout 120 python tools/attn_probe.py 2>/dev/null | grep -v amdgpu
