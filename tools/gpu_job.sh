#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git); the committed form at round end is the round-end check: full GPU suite, smoke, default bench line -> gpurun_out/TAG/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
# r19b: the fp32-stream mode on the fused kernels (ffn LayerNorm on fp32 X, out16 -> fused temporal block): parity, then A/B
timeout 1500 python -m pytest tests/test_hip_ffn.py tests/test_hip_res32.py tests/test_hip_temporal_block.py -q -m gpu -rP -x > $O/${TAG}_tests.log 2>&1
grep -n "passed\|failed\|rror" $O/${TAG}_tests.log | tail -5; grep "\[tolerance\]" $O/${TAG}_tests.log | cut -c1-330
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
for cfg in "A:--dtype fp16:" "B:--dtype fp16 --residual-fp32:" "C:--dtype fp16 --residual-fp32:WIW_TEMPORAL_UNFUSED_RES32=1" "D:--dtype fp16 --residual-fp32:WIW_FFN_NO_LN_RES32=1" "E:--dtype fp16 --residual-fp32:WIW_TEMPORAL_UNFUSED_RES32=1 WIW_FFN_NO_LN_RES32=1"; do
  IFS=: read name flags envs <<< "$cfg"
  env $envs timeout 400 $B $flags 2>/dev/null | tail -1 > $O/${TAG}_bench_$name.json
  python - "$O/${TAG}_bench_$name.json" "$name $flags $envs" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
ok = d.get("other_kernels", {})
print(sys.argv[2], "->", d["value"], "frames/s", d["ms_per_step"], "ms |", " ".join(f"{k}:{v['seconds']*1e3/5:.2f}" for k, v in ok.items()), "| dense", round(d["gemm_kernels"]["gemm_kernel<dense>"]["seconds"]*1e3/5, 2))
PY
done
