#!/usr/bin/env python
"""Turn a rocprofv3 results database (rocpd sqlite, `rocprofv3 --kernel-trace --stats -d DIR -o NAME`)
into the small text summaries kept under profiles/:  per-kernel calls / total / average / share, and
(optionally, with --pmc) the per-kernel sums of collected counters.

    python tools/rocprof_summary.py gpurun_out/prof/NAME_results.db profiles/r01_kernel_stats.csv
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:<>, ]+?)\(", name)
    base = m.group(1) if m else name
    if base.startswith("at::native"):
        k = re.search(r"(\w+_kernel\w*|\w+Functor\w*|CatArrayBatchedCopy\w*|roll_cuda_kernel)", name)
        base = "torch:" + (k.group(1) if k else base[:40])
    return base[:80]


def main():
    db, out = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                          "from kernels group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
        for name, n, tot, avg, mn, mx in rows:
            f.write(f"\"{short(name)}\",{n},{tot / 1e3:.1f},{avg / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100 * tot / total:.2f}\n")
    print(f"wrote {out}: {len(rows)} kernels, {total / 1e6:.1f} ms of kernel time")
    if "--pmc" in sys.argv:
        try:
            q = ("select k.name, p.name, sum(e.value), count(*) from pmc_events e join kernels k on e.event_id = k.id "
                 "join pmc_info p on e.pmc_id = p.id group by k.name, p.name")
            with open(out.replace(".csv", "_pmc.csv"), "w") as f:
                f.write("kernel,counter,sum,samples\n")
                for kn, pn, v, n in c.execute(q):
                    f.write(f"\"{short(kn)}\",{pn},{v},{n}\n")
        except sqlite3.Error as e:
            print("pmc tables not readable:", e)


if __name__ == "__main__":
    main()
