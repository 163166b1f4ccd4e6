# round 2, GPU call B: suite on the new server / parallel code, then profiles (kernel stats, PMC) and bench with per-family events
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s --timeout 600 > $O/t_all.log 2>&1; echo "suite rc=$?" | tee -a $O/summary.txt
tail -4 $O/t_all.log
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o ks -- python bench.py --num-inference-steps 2 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $O/ks.log 2>&1
python tools/rocprof_summary.py $(find /tmp/prof_ks -name "*results.db" | head -1) $O/r02b_kernel_stats_2step.csv
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc$i -o p$i -- python bench.py --num-inference-steps 1 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-events > $O/pmc$i.log 2>&1
done
python tools/pmc_summary.py $O/r02b_pmc_1step.csv $(find /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 -name "*results.db")
timeout 300 python bench.py --dump-shapes $O/r02b_shapes.txt 2>/dev/null | tail -1 > $O/r02b_bench.json; cut -c1-400 $O/r02b_bench.json
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --end-to-end 2>/dev/null | tail -1 > $O/r02b_bench_e2e.json; python -c "
import json; d=json.load(open('$O/r02b_bench_e2e.json')); print(d['value'], d.get('end_to_end'))"
head -12 $O/r02b_kernel_stats_2step.csv; grep temporal $O/r02b_pmc_1step.csv
