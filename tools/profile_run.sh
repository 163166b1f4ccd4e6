# rocprofv3 kernel-stats (2-step) + PMC (1-step, three separate passes) of bench.py; summaries -> gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
mkdir -p gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o ks -- python bench.py --num-inference-steps 2 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > gpurun_out/${TAG}_ks.log 2>&1
python tools/rocprof_summary.py $(find /tmp/prof_ks -name "*results.db" | head -1) gpurun_out/${TAG}_kernel_stats_2step.csv
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc$i -o p$i -- python bench.py --num-inference-steps 1 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events > gpurun_out/${TAG}_pmc$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/${TAG}_pmc_1step.csv $(find /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 -name "*results.db")
timeout 200 python bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench.json
cut -c1-160 gpurun_out/${TAG}_bench.json
