#!/usr/bin/env python
"""Print per-kernel PMC counter means from a rocprofv3 rocpd database (view `counters_collection`)."""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = c.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection "
                 "group by kernel_name, counter_name order by kernel_name")
for kn, pn, v, n, d in rows:
    kn = re.sub(r"\(anonymous namespace\)::", "", kn)[:58]
    if pat in kn:
        print(f"{kn:58s} {pn:28s} {v:18.1f} n={n} avg_dur_us={(d or 0) / 1e3:.1f}")
