#!/usr/bin/env python
"""Round 6: joules per TFLOP of `v_mfma_f32_16x16x32` against `v_mfma_f32_32x32x16` under the 256x320 GEMM
kernel's own LDS read pattern and accumulator footprint (tools/ubench/mfma_energy.hip), sampled with rocm-smi while each
variant is held for SECONDS (default 6).  The binary is built in the build container into tools/ablate/ (git-ignored, travels
with gpurun); it is rebuilt here if missing.

    python tools/mfma_energy.py        -> one line per variant: TFLOP/s, W, MHz, J/TFLOP"""
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "ablate", "mfma_energy")
SRC = os.path.join(ROOT, "tools", "ubench", "mfma_energy.hip")


def smi():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=10).stdout
        r = [x for x in out.splitlines() if x.startswith("card")][0]
        clocks = [int(x) for x in re.findall(r"\((\d+)Mhz\)", r)]
        return max(clocks[2:4]), float(r.split(",")[-1])
    except Exception:
        return None


def main():
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(BIN), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-w", SRC, "-o", BIN])
    sec = float(os.environ.get("SECONDS", "6"))
    order = [int(v) for v in os.environ.get("VARIANTS", "0,1,2,3,4,5,0,1").split(",")]
    print(f"# {time.strftime('%F %T')}  each variant held {sec:.0f} s; idle sample: {smi()}")
    for v in order:
        samples, stop = [], [False]

        def loop():
            time.sleep(1.0)                 # let the clock settle under the load
            while not stop[0]:
                s = smi()
                if s:
                    samples.append(s)
                time.sleep(0.15)

        th = threading.Thread(target=loop)
        th.start()
        out = subprocess.run([BIN, str(v), str(sec)], capture_output=True, text=True).stdout.strip()
        stop[0] = True
        th.join()
        m = re.search(r"([\d.]+) TFLOP/s", out)
        busy = samples[:-1] if len(samples) > 3 else samples
        if m and busy:
            tf = float(m.group(1))
            w = sum(s[1] for s in busy) / len(busy)
            f = sum(s[0] for s in busy) / len(busy)
            print(f"{out}\n    -> {w:6.0f} W  {f:5.0f} MHz  {w / tf:6.3f} J/TFLOP  ({tf / (f / 2400.0 * 2500.0):.3f} of the dense peak at that clock; {len(busy)} samples)", flush=True)
        else:
            print(out, "(no power samples)", flush=True)
        time.sleep(2.0)


if __name__ == "__main__":
    sys.exit(main())
