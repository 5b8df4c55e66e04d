#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table --
the same columns `rocprofv3 --stats` prints (calls, total, average, percentage)."""
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}"]
    for n, c, s, a, mn, mx in rows:
        short = n if len(n) <= 90 else n[:87] + "..."
        lines.append(f"{short:90s} {c:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}")
    lines.append(f"{'TOTAL kernel time':90s} {sum(r[1] for r in rows):7d} {tot/1e6:10.3f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
