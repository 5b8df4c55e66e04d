#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database (view counters_collection)."""
import sqlite3
import sys


def main(db, out=None, filt=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                       "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    lines = []
    last = None
    for k, c, n, v, d in rows:
        if filt and filt not in k:
            continue
        if k != last:
            lines.append(f"\n{k[:140]}   (dispatches sampled: {n}, avg duration {d/1e3 if d else 0:.1f} us)")
            last = k
        lines.append(f"    {c:32s} {v:18.1f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None, sys.argv[3] if len(sys.argv) > 3 else None)
