cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r10m; mkdir -p $O
echo "== res32 tests"; timeout 1200 python -m pytest tests/test_hip_res32.py -q -x -rP > $O/r10m_res32_tests.log 2>&1; grep -n "256x320\|passed\|failed\|rror\|\[tolerance\]\|rms_rel=" $O/r10m_res32_tests.log | cut -c1-200 | tail -40
echo "== bench res32"; timeout 600 python bench.py --dtype fp16 --residual-fp32 --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/r10m_bench_fp16_res32.json; python -c "
import json; d=json.loads(open('$O/r10m_bench_fp16_res32.json').read()); print(d['value'], d['gemm_kernels'])"
echo "== bench res32 (256x160 F32E kernels)"; WIW_GEMM_HUGE_NO_F32=1 WIW_CONV_NO_HALO=1 timeout 600 python bench.py --dtype fp16 --residual-fp32 --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['gemm_kernels'])"
