#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of this ROCm's `rocprofv3 --kernel-trace`): per-kernel totals
(the `--stats` table) or the timeline of the last complete step between two occurrences of a marker kernel.

    python tools/rocpd_summary.py results.db [--top 25] [--last-steps N --marker NAME]
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"^void ", "", n).replace("cti::(anonymous namespace)::", "").replace("cti::", "")
    n = re.sub(r"Geo<([^>]*)>", lambda m: "G<" + m.group(1).replace(" ", "") + ">", n)
    return re.sub(r"\(.*", "", n)[:78]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    skip = float(sys.argv[sys.argv.index("--skip-frac") + 1]) if "--skip-frac" in sys.argv else 0.0
    rows = rows[int(len(rows) * skip):]
    tot = defaultdict(lambda: [0, 0.0])
    for n, s, e in rows:
        t = tot[short(n)]
        t[0] += 1
        t[1] += (e - s) / 1e3
    busy = sum(v[1] for v in tot.values())
    span = (rows[-1][2] - rows[0][1]) / 1e3
    print("kernels %d  busy %.1f us  span %.1f us  (%.0f%% busy)" % (len(rows), busy, span, 100 * busy / span))
    print("%-80s %8s %12s %10s %6s" % ("name", "calls", "total_us", "avg_us", "%"))
    for k, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-80s %8d %12.1f %10.2f %6.1f" % (k, c, us, us / c, 100 * us / busy))


if __name__ == "__main__":
    main()
