cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_clip.py -m gpu -q -s --timeout 300 > $O/t_clip.log 2>&1; echo "clip rc=$?" | tee -a $O/summary.txt
grep -E "parity.*CLIP|passed|failed|Error" $O/t_clip.log | cut -c1-200 | tail -8
timeout 300 python -m pytest tests/test_hip_kernels.py tests/test_hip_served_width.py tests/test_hip_fullsize.py tests/test_hip_vae.py -m gpu -q --timeout 300 > $O/t_gn.log 2>&1; echo "kernels rc=$?" | tee -a $O/summary.txt; tail -2 $O/t_gn.log
timeout 120 python tools/norm_probe.py 2>/dev/null | grep gn_stats
