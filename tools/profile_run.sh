# rocprofv3 kernel-stats (2-step) + PMC (1-step, three separate passes: never combined with tracing domains other than
# kernel-trace) of bench.py, the default bench line with the per-shape dump, and the B = 8 / end-to-end line -> gpurun_out/TAG/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r18}
O=gpurun_out/$TAG; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o ks -- python bench.py --num-inference-steps 2 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-graph --no-calibrate --no-power > $O/ks.log 2>&1
python tools/rocprof_summary.py $(find /tmp/prof_ks -name "*results.db" | head -1) $O/${TAG}_kernel_stats_2step.csv
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc$i -o p$i -- python bench.py --num-inference-steps 1 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events --no-graph --no-calibrate --no-power > $O/pmc$i.log 2>&1
done
python tools/pmc_summary.py $O/${TAG}_pmc_1step.csv $(find /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 -name "*results.db")
# the temporal block alone (MFMA pipe of temporal_block_resident_kernel (C = 320) / temporal_block_kernel at the level shapes; PMC in its own pass)
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_t -o pt -- env ONLY_FUSED=1 ITERS=5 python tools/temporal_probe.py > $O/pmc_temporal.log 2>&1
python tools/pmc_summary.py $O/${TAG}_pmc_temporal_probe.csv $(find /tmp/pmc_t -name "*results.db")
# the default line (what the driver runs: `box` calibration, `extra` legs = B = 8, fp16, fp16 + fp32 stream, 2 training steps)
timeout 600 python bench.py --dump-shapes $O/${TAG}_shapes.txt 2>/dev/null | tail -1 > $O/${TAG}_bench.json; cut -c1-300 $O/${TAG}_bench.json
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --end-to-end 2>/dev/null | tail -1 > $O/${TAG}_bench_e2e.json
grep temporal $O/${TAG}_pmc_1step.csv $O/${TAG}_pmc_temporal_probe.csv
# the <= 1e-3 configuration with per-family rows (fp16 + fp32 residual stream)
timeout 300 python bench.py --dtype fp16 --residual-fp32 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_fp16_res32.json; cut -c1-200 $O/${TAG}_bench_fp16_res32.json
