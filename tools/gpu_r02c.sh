cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_clip.py tests/test_hip_server.py tests/test_hip_vae.py -m gpu -q -s --timeout 300 > $O/t_clip.log 2>&1; echo "clip/server rc=$?" | tee -a $O/summary.txt
grep -E "parity|passed|failed|Error" $O/t_clip.log | cut -c1-200 | tail -25
timeout 300 python -m pytest tests/test_hip_kernels.py tests/test_hip_served_width.py -m gpu -q -k "groupnorm" --timeout 300 > $O/t_gn.log 2>&1; echo "gn rc=$?" | tee -a $O/summary.txt; tail -2 $O/t_gn.log
timeout 120 python tools/norm_probe.py > $O/norm_probe.log 2>&1; cat $O/norm_probe.log
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --end-to-end 2>/dev/null | tail -1 > $O/bench_e2e.json; python -c "
import json; d=json.load(open('$O/bench_e2e.json')); print(d['value'], d.get('end_to_end'), d['other_kernels']['groupnorm'])"
