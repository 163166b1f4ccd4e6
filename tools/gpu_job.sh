#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git).  This form: the free-running K loop of gemm_huge.hip (PIPE, round 5) against the eight-slot loop (-DWIW_HUGE_PIPE=0).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
L=$O/${TAG}_pipe_ab.txt
NP=tools/ablate/libwiw_nopipe.so; D0=tools/ablate/libwiw_dma0.so
SH="64512,5120,640,0,1 16128,10240,1280,0,1 64512,640,2560 16128,1280,5120 258048,320,320 64512,640,640 16128,1280,1280 258048,960,320 64512,1920,640 258048,320,960,4 64512,640,1920,4 16128,1280,3840,4"
{
echo "# bit-level A/B (tools/pipe_check.py): PIPE, PIPE with WIW_PIPE_DMA=0, eight-slot loop"
timeout 300 python tools/pipe_check.py 2>/dev/null > $O/chk_pipe.txt; WIW_LIB=$D0 timeout 300 python tools/pipe_check.py 2>/dev/null > $O/chk_dma0.txt; WIW_LIB=$NP timeout 300 python tools/pipe_check.py 2>/dev/null > $O/chk_slots.txt
echo "lines: $(wc -l < $O/chk_slots.txt); PIPE vs slots differing lines: $(diff $O/chk_pipe.txt $O/chk_slots.txt | grep -c '^<'); DMA=0 vs slots: $(diff $O/chk_dma0.txt $O/chk_slots.txt | grep -c '^<')"
echo "# gemm_probe (TILED=1), interleaved"
for r in 1 2; do
echo "## PIPE (K tile kt + 2 behind the barrier)";    TILED=1 timeout 300 python tools/gemm_probe.py $SH 2>&1 | grep "M="
echo "## PIPE, WIW_PIPE_DMA=0 (K tile kt + 1 during tile kt)";    WIW_LIB=$D0 TILED=1 timeout 300 python tools/gemm_probe.py $SH 2>&1 | grep "M="
echo "## eight slots";   WIW_LIB=$NP TILED=1 timeout 300 python tools/gemm_probe.py $SH 2>&1 | grep "M="
done
} > $L 2>&1
cat $L
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_served_width.py tests/test_hip_unet.py -q -m gpu -x > $O/${TAG}_tests.log 2>&1
grep -n "passed\|failed\|rror" $O/${TAG}_tests.log | tail -6
