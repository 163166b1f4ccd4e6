#!/bin/bash
# same-box matrix: round-1 build vs this build, with and without per-launch events, and this build with each round-2 change knobbed off
mkdir -p gpurun_out/r02s
run() { # name, dir, env..., args
  local name=$1; shift; local dir=$1; shift
  (cd $dir && env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $EXTRA 2>/dev/null | tail -1) > gpurun_out/r02s/$name.json
}
R1=tools/ablate/r01_tree
EXTRA="" run r01_ev $R1 X=1
EXTRA="" run r02_ev . X=1
EXTRA="--no-kernel-events" run r01_noev $R1 X=1
EXTRA="--no-kernel-events" run r02_noev . X=1
EXTRA="--no-kernel-events" run r02_untiled . WIW_W_UNTILED=1
EXTRA="--no-kernel-events" run r02_tunfused . WIW_TEMPORAL_UNFUSED=1
EXTRA="--no-kernel-events" run r02_swappedvt . WIW_SWAPPED_VT=1
EXTRA="--no-kernel-events" run r02_nosplitk . WIW_NO_SPLITK=1
EXTRA="--no-kernel-events" run r01_noev2 $R1 X=1
EXTRA="--no-kernel-events" run r02_noev2 . X=1
python - <<'PY'
import json,glob
for n in ("r01_ev","r02_ev","r01_noev","r02_noev","r02_untiled","r02_tunfused","r02_swappedvt","r02_nosplitk","r01_noev2","r02_noev2"):
    try:
        d=json.loads(open(f"gpurun_out/r02s/{n}.json").read())
        print(f"{n:14s} {d['value']:.4f} frames/s  {d['ms_per_step']:.1f} ms")
    except Exception as e: print(n, "ERR", e)
PY
