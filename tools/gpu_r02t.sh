#!/bin/bash
# after the SK template split + sampled events: kernel tests, default bench line (x2), shapes dump, B=8
mkdir -p gpurun_out/r02t
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_served_width.py -m gpu -q > gpurun_out/r02t/tests.log 2>&1; tail -2 gpurun_out/r02t/tests.log
timeout 300 python bench.py --dump-shapes gpurun_out/r02t/r02t_shapes.txt 2>/dev/null | tail -1 > gpurun_out/r02t/r02t_bench.json
timeout 300 python bench.py --no-cpu-baseline --event-every 1 2>/dev/null | tail -1 > gpurun_out/r02t/r02t_bench_allevents.json
timeout 300 python bench.py --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 > gpurun_out/r02t/r02t_bench_noevents.json
(cd tools/ablate/r01_tree && timeout 300 python bench.py --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1) > gpurun_out/r02t/r01_bench_noevents.json
timeout 400 python bench.py --batch 8 --steps 1 --warmup 1 --no-cpu-baseline --end-to-end 2>/dev/null | tail -1 > gpurun_out/r02t/r02t_bench_batch8.json
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --end-to-end 2>/dev/null | tail -1 > gpurun_out/r02t/r02t_bench_e2e.json
python - <<'PY'
import json
for n in ("r02t_bench","r02t_bench_allevents","r02t_bench_noevents","r01_bench_noevents","r02t_bench_batch8","r02t_bench_e2e"):
    try:
        d=json.loads(open(f"gpurun_out/r02t/{n}.json").read())
        print(f"{n:26s} {d['value']:.4f} frames/s {d['ms_per_step']:.1f} ms util={d.get('mfma_util')} roof={d.get('roofline',{}).get('frac')} e2e={d.get('end_to_end',{}).get('frames_per_s')}")
    except Exception as e: print(n, "ERR", e)
PY
