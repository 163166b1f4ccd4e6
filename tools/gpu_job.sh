#!/bin/bash
# Per-call GPU job: `gpurun -- 'bash tools/gpu_job.sh TAG'`.  This file is edited for every experiment (its history is in
# git).  This form: the fitted GELU (round 5) against the A&S 7.1.26 build of the same sources, on one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-job}
O=gpurun_out/$TAG; mkdir -p $O
L=$O/${TAG}_gelu_ab.txt
AS=tools/ablate/libwiw_as7126.so; NG=tools/ablate/libwiw_noge.so
{
echo "# fitted sigmoid-form GELU (default build) vs -DWIW_GELU_AS7126=1 (Abramowitz-Stegun 7.1.26, 14 instructions) vs no GELU math at all (FFN_ABLATE=1, wrong results)"
for r in 1 2; do
echo "## ffn_probe default";   ONLY_FUSED=1 timeout 300 python tools/ffn_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
echo "## ffn_probe as7126";    WIW_LIB=$AS ONLY_FUSED=1 timeout 300 python tools/ffn_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
done
echo "## ffn_probe noge";      WIW_LIB=$NG ONLY_FUSED=1 timeout 300 python tools/ffn_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
for r in 1 2; do
echo "## gemm_probe GEGLU default"; TILED=1 timeout 300 python tools/gemm_probe.py 64512,5120,640,0,1 16128,10240,1280,0,1 4032,10240,1280,0,1 2>&1 | grep "M="
echo "## gemm_probe GEGLU as7126";  WIW_LIB=$AS TILED=1 timeout 300 python tools/gemm_probe.py 64512,5120,640,0,1 16128,10240,1280,0,1 4032,10240,1280,0,1 2>&1 | grep "M="
done
} > $L 2>&1
cat $L
timeout 1500 python -m pytest tests/test_hip_ffn.py tests/test_hip_kernels.py tests/test_hip_unet.py tests/test_hip_res32.py tests/test_hip_clip.py tests/test_hip_fp16.py tests/test_hip_served_width.py -q -m gpu -rP -x > $O/${TAG}_tests.log 2>&1
grep -n "passed\|failed\|rror" $O/${TAG}_tests.log | tail -6
echo "== bench default"; timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${TAG}_bench.json; cut -c1-200 $O/${TAG}_bench.json
echo "== bench as7126"; WIW_LIB=$AS timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${TAG}_bench_as7126.json; cut -c1-200 $O/${TAG}_bench_as7126.json
python - $TAG <<'PY'
import json,sys
for n in ("bench","bench_as7126"):
    j=json.load(open(f"gpurun_out/%s/%s_%s.json"%(sys.argv[1] if len(sys.argv)>1 else "r15a", sys.argv[1] if len(sys.argv)>1 else "r15a", n)))
    print(n, j["value"], {k:(v["seconds"], v.get("tflops")) for k,v in j["other_kernels"].items() if k in ("ffn_fused",)}, j["roofline"]["achieved"])
PY
