#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table --
the same columns `rocprofv3 --stats` prints (calls, total, average, percentage)."""
import sqlite3
import sys


def main(db, out=None, by_grid=False):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    key = name_col
    if by_grid:      # --by-grid: one row per (kernel, grid): the same template instantiation at different problem shapes
        gcols = [c for c in cols if "grid" in c.lower()]
        if gcols:
            key = name_col + " || ' g=' || " + " || 'x' || ".join(f"cast({c} as text)" for c in gcols)
        else:
            print("(no grid columns in the kernels view: " + ", ".join(cols) + ")")
    rows = cur.execute(f"select {key}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by 1 order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}"]
    for n, c, s, a, mn, mx in rows:
        if by_grid and " g=" in n and len(n) > 90:
            nm, gr = n.rsplit(" g=", 1)
            n = nm[: 86 - len(gr)] + " g=" + gr
        short = n if len(n) <= 90 else n[:87] + "..."
        lines.append(f"{short:90s} {c:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}")
    lines.append(f"{'TOTAL kernel time':90s} {sum(r[1] for r in rows):7d} {tot/1e6:10.3f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--by-grid"]
    main(args[0], args[1] if len(args) > 1 else None, by_grid="--by-grid" in sys.argv)
