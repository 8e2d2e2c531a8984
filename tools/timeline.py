#!/usr/bin/env python
"""Compact per-dispatch timeline from a rocprofv3 `--kernel-trace --output-format csv` run: the last `--last` dispatches as
`start_us,end_us,queue,short kernel name` (relative to the first kept dispatch), for offline critical-path analysis of the
segment schedule (which stream is busy when, where the main stream waits)."""
import argparse
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from knames import tg_name


def short(n):
    t = tg_name(n)
    if t:
        return t
    n = n.replace("void ", "")
    return n.split("(")[0][:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("out")
    ap.add_argument("--last", type=int, default=13000)
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no kernel_trace.csv under %s" % a.dir)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"),
                             r["Kernel_Name"]))
    rows.sort()
    rows = rows[-a.last:]
    t0 = rows[0][0]
    with open(a.out, "w") as o:
        o.write("start_us,end_us,queue,stream,kernel\n")
        for s, e, q, st, n in rows:
            o.write("%.2f,%.2f,%s,%s,%s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, q, st, short(n).replace(",", ";")))
    print("wrote %d dispatches to %s" % (len(rows), a.out))


if __name__ == "__main__":
    main()
