#!/usr/bin/env python
"""Summarise a rocprofv3 `--kernel-trace --stats` result database (rocpd sqlite) as a per-kernel table."""
import sqlite3
import sys


def main(db, out=None, top=40):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary of %s" % db,
             "# total kernel time %.3f ms over %d dispatches" % (tot / 1e3, sum(r[1] for r in rows)),
             "%8s %12s %10s %7s  %s" % ("calls", "total_us", "avg_us", "pct", "kernel")]
    for name, calls, total, avg, pct in rows[:top]:
        if len(name) > 150:
            name = name[:147] + "..."
        lines.append("%8d %12.1f %10.3f %6.2f%%  %s" % (calls, total, avg, pct, name))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, int(sys.argv[3]) if len(sys.argv) > 3 else 40)
