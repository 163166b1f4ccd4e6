# round 2, GPU call A: new tests first (fused temporal block under a short timeout), then the suite, probes, bench A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_temporal_block.py -m gpu -q -s --timeout 120 > $O/t_temporal.log 2>&1; echo "temporal rc=$?" | tee -a $O/summary.txt
tail -5 $O/t_temporal.log
timeout 900 python -m pytest tests -m gpu -q -s --timeout 600 --deselect tests/test_hip_temporal_block.py > $O/t_all.log 2>&1; echo "suite rc=$?" | tee -a $O/summary.txt
tail -8 $O/t_all.log
timeout 200 python tools/temporal_probe.py > $O/temporal_probe.log 2>&1; cat $O/temporal_probe.log
for tile in big small; do
  echo "== WIW_GEMM_TILE=$tile" >> $O/gemm_probe.log
  WIW_GEMM_TILE=$tile timeout 120 python tools/gemm_probe.py 258048,2560,320,0,1 258048,320,320 258048,640,320 258048,960,320 64512,5120,640,0,1 >> $O/gemm_probe.log 2>&1
done
RES=1 WIW_GEMM_TILE=small timeout 100 python tools/gemm_probe.py 258048,320,320 258048,320,1280 >> $O/gemm_probe.log 2>&1
RES=1 timeout 100 python tools/gemm_probe.py 258048,320,320 258048,320,1280 >> $O/gemm_probe.log 2>&1
cat $O/gemm_probe.log
timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_fused.json; cut -c1-200 $O/bench_fused.json
WIW_TEMPORAL_UNFUSED=1 timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_unfused.json; cut -c1-200 $O/bench_unfused.json
