cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_kernels.py tests/test_hip_temporal_block.py -m gpu -q --timeout 300 2>&1 | tail -4
for t in 0 1; do
  echo "== TILED=$t"; if [ $t = 1 ]; then export TILED=1; fi
  timeout 200 python tools/gemm_probe.py 258048,2560,320,0,1 258048,320,320 258048,640,320 64512,5120,640,0,1 64512,640,2560 258048,320,1280 258048,320,2880,1 258048,320,8640,1 64512,640,5760,1 16128,1280,11520,1 258048,320,960,4 2>/dev/null | grep -v amdgpu
done
unset TILED
timeout 100 python tools/temporal_probe.py 2>/dev/null | grep -v amdgpu
timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120
WIW_W_UNTILED=1 timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120
