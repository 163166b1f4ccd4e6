#!/usr/bin/env python
"""Per-kernel PMC summary from rocprofv3 rocpd databases -> CSV for profiles/.

    python tools/pmc_summary.py out.csv db1 [db2 ...]
Sums each counter over all dispatches of a kernel and reports per-launch means.  For FETCH_SIZE /
WRITE_SIZE (KiB units) the derived HBM bytes follow MI355X_MICROARCH.md §HBM: bytes = KiB * 1024 and,
on gfx950, FETCH_SIZE under-reports wide coalesced reads by exactly 2x -> `fetch_bytes_corrected`.
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    base = m.group(1) if m else name
    return ("torch:" + base[-40:]) if base.startswith("at::") else base[:60]


def main():
    out, dbs = sys.argv[1], sys.argv[2:]
    agg = {}
    for db in dbs:
        c = sqlite3.connect(db)
        for kn, cn, tot, n, dur in c.execute("select kernel_name, counter_name, sum(value), count(*), sum(duration) "
                                             "from counters_collection group by kernel_name, counter_name"):
            d = agg.setdefault(short(kn), {})
            d[cn] = (tot, n, dur)
    names = sorted({cn for d in agg.values() for cn in d})
    with open(out, "w") as f:
        f.write("kernel,launches,total_us," + ",".join(f"{n}_per_launch" for n in names) + ",derived\n")
        for k, d in sorted(agg.items(), key=lambda kv: -max(v[2] for v in kv[1].values())):
            n = max(v[1] for v in d.values())
            dur = max(v[2] for v in d.values())
            cols = [f"{d[c][0] / d[c][1]:.1f}" if c in d else "" for c in names]
            der = []
            if "FETCH_SIZE" in d:
                der.append(f"fetch_bytes_corrected={2 * 1024 * d['FETCH_SIZE'][0] / d['FETCH_SIZE'][1]:.3e}")
            if "WRITE_SIZE" in d:
                der.append(f"write_bytes={1024 * d['WRITE_SIZE'][0] / d['WRITE_SIZE'][1]:.3e}")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
                # busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
                der.append(f"mfma_busy_frac={8 * d['SQ_VALU_MFMA_BUSY_CYCLES'][0] / (d['GRBM_GUI_ACTIVE'][0] * 1024):.3f}")
            f.write(f"\"{k}\",{n},{dur / 1e3:.1f}," + ",".join(cols) + ",\"" + " ".join(der) + "\"\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
