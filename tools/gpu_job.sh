#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git).  This form: the free-running K loop with the halo-staged convolutions (WIW_HUGE_PIPE=2) vs without (=1) vs slots (=0).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
L=$O/${TAG}_pipe_halo_ab.txt
NP=tools/ablate/libwiw_nopipe.so; P1=tools/ablate/libwiw_pipe1.so
{
echo "# bit-level A/B (tools/pipe_check.py): PIPE incl. halo convolutions vs eight-slot loop"
timeout 400 python tools/pipe_check.py 2>/dev/null > $O/chk_pipe.txt; WIW_LIB=$NP timeout 400 python tools/pipe_check.py 2>/dev/null > $O/chk_slots.txt
echo "lines: $(wc -l < $O/chk_slots.txt); differing lines: $(diff $O/chk_pipe.txt $O/chk_slots.txt | grep -c '^<'); non-finite: $(grep -c finite=False $O/chk_pipe.txt)"
diff $O/chk_pipe.txt $O/chk_slots.txt | head -8
echo "# sustained rate, power, clock, J/TF (tools/power_by_kernel.py)"
echo "## WIW_HUGE_PIPE=2 (default build)"; SECONDS=3 timeout 400 python tools/power_by_kernel.py 2>&1 | grep "conv3x3 halo\|dense GEGLU\|conv_t3"
echo "## WIW_HUGE_PIPE=0"; WIW_LIB=$NP SECONDS=3 timeout 400 python tools/power_by_kernel.py 2>&1 | grep "conv3x3 halo\|dense GEGLU\|conv_t3"
} > $L 2>&1
cat $L
echo "== bench PIPE=2"; timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${TAG}_bench.json; cut -c1-130 $O/${TAG}_bench.json
echo "== bench PIPE=1"; WIW_LIB=$P1 timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${TAG}_bench_pipe1.json; cut -c1-130 $O/${TAG}_bench_pipe1.json
echo "== bench slots"; WIW_LIB=$NP timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${TAG}_bench_slots.json; cut -c1-130 $O/${TAG}_bench_slots.json
