#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git); the committed form at round end is the round-end check: full GPU suite, smoke, default bench line -> gpurun_out/TAG/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
# r19c: the cooperative one-read GroupNorm (ABI 16): parity / bit-identity first, then the same-box A/B
timeout 300 python -m pytest tests/test_hip_kernels.py -q -m gpu -rP -x -k "groupnorm" > $O/${TAG}_gn_tests.log 2>&1
grep -n "passed\|failed\|rror" $O/${TAG}_gn_tests.log | tail -5
timeout 300 python -m pytest tests/test_hip_res32.py -q -m gpu -x -k "cooperative or kernels_on_fp32 or groupnorm" 2>&1 | tail -3
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
for cfg in "A::" "B::WIW_GN_COOP=0" "C:--dtype fp16 --residual-fp32:" "D:--dtype fp16 --residual-fp32:WIW_GN_COOP=0" "E:--dtype fp16:"; do
  IFS=: read name flags envs <<< "$cfg"
  env $envs timeout 400 $B $flags 2>/dev/null | tail -1 > $O/${TAG}_bench_$name.json
  python - "$O/${TAG}_bench_$name.json" "$name $flags $envs" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
ok = d.get("other_kernels", {})
print(sys.argv[2], "->", d["value"], "frames/s", d["ms_per_step"], "ms |", " ".join(f"{k}:{v['seconds']*1e3/5:.2f}" for k, v in ok.items()), "| dense", round(d["gemm_kernels"]["gemm_kernel<dense>"]["seconds"]*1e3/5, 2))
PY
done
