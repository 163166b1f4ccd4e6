#!/bin/bash
mkdir -p gpurun_out/r02v
timeout 900 python -m pytest tests/test_hip_server.py -m gpu -q -s -k "end_to_end" > gpurun_out/r02v/server.log 2>&1
grep -E "parity|passed|failed|rror" gpurun_out/r02v/server.log | head
timeout 300 python bench.py --no-cpu-baseline --dtype fp16 2>gpurun_out/r02v/fp16.err | tail -1 > gpurun_out/r02v/r02v_bench_fp16.json
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02v/r02v_bench_bf16.json
timeout 300 python bench.py --no-cpu-baseline --dtype fp16 --steps 1 --warmup 1 --end-to-end 2>/dev/null | tail -1 > gpurun_out/r02v/r02v_bench_fp16_e2e.json
tail -3 gpurun_out/r02v/fp16.err
python - <<'PY'
import json
for n in ("r02v_bench_fp16","r02v_bench_bf16","r02v_bench_fp16_e2e"):
    try:
        d=json.loads(open(f"gpurun_out/r02v/{n}.json").read())
        print(f"{n:22s} {d['dtype']} {d['value']:.4f} frames/s {d['ms_per_step']:.1f} ms util={d.get('mfma_util')} roof={d.get('roofline',{}).get('frac')} e2e={d.get('end_to_end',{}).get('frames_per_s')}")
    except Exception as e: print(n, "ERR", e)
PY
