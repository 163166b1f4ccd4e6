cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc$i -o p$i -- python bench.py --num-inference-steps 1 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events > gpurun_out/pmc/log$i.txt 2>&1
done
python tools/pmc_summary.py gpurun_out/r01e_pmc_1step.csv $(find /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 -name "*results.db")
head -5 gpurun_out/r01e_pmc_1step.csv | cut -c1-300
