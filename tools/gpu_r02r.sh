#!/bin/bash
# same-box A/B: round-1 final build (tools/ablate/r01_tree, commit eaeac7c) vs this build, interleaved
mkdir -p gpurun_out/r02r
for i in 1 2; do
  (cd tools/ablate/r01_tree && timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/r02r/r01_$i.json
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02r/r02_$i.json
done
python - <<'PY'
import json
for n in ("r01_1","r02_1","r01_2","r02_2"):
    try:
        d=json.loads(open(f"gpurun_out/r02r/{n}.json").read())
        print(n, d["value"], d["ms_per_step"], d.get("mfma_util", d.get("mfma_util_algorithmic")), d["roofline"]["achieved"])
    except Exception as e: print(n, "ERR", e)
PY
