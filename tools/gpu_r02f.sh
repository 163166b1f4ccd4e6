cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/ubench/lds_fill.hip -o /tmp/lds_fill && timeout 120 /tmp/lds_fill | tee $O/lds_fill.log
timeout 60 python tools/norm_probe.py 2>/dev/null | grep gn_stats | head -4
