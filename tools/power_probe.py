#!/usr/bin/env python
"""Sample the GPU's shader clock and package power with rocm-smi (a few Hz) while a command runs and print the means over the
busy samples: is a rollout running against the power management of the board?  (Round 5: it is — 2.06-2.12 GHz at ~1270 W of a
1400 W cap during the denoising loop, where a pure MFMA loop on register operands holds 2.4 GHz.)

    python tools/power_probe.py -- python bench.py --no-cpu-baseline --no-extras --no-kernel-events"""
import re
import subprocess
import sys
import threading


def sample():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=10).stdout
    except Exception:
        return None
    rows = [r for r in out.splitlines() if r.startswith("card")]
    if not rows:
        return None
    r = rows[0]
    clocks = [int(x) for x in re.findall(r"\((\d+)Mhz\)", r)]
    try:
        power = float(r.split(",")[-1])
    except ValueError:
        return None
    return (max(clocks[2:4]) if len(clocks) >= 4 else max(clocks), power)     # sclk is the third clock column


def main():
    cmd = sys.argv[sys.argv.index("--") + 1:]
    samples = []
    stop = [False]

    def loop():
        while not stop[0]:
            s = sample()
            if s:
                samples.append(s)

    th = threading.Thread(target=loop)
    th.start()
    rc = subprocess.call(cmd)
    stop[0] = True
    th.join()
    if samples:
        pmax = max(s[1] for s in samples)
        busy = [s for s in samples if s[1] > 0.8 * pmax]
        mean = lambda xs: sum(xs) / len(xs)
        print(f"[power_probe] {len(samples)} samples, {len(busy)} with power > 80 % of the maximum seen ({pmax:.0f} W): "
              f"sclk mean {mean([s[0] for s in busy]):.0f} MHz (min {min(s[0] for s in busy)}, max {max(s[0] for s in busy)}), power mean {mean([s[1] for s in busy]):.0f} W")
    return rc


if __name__ == "__main__":
    sys.exit(main())
