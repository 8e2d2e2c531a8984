#!/usr/bin/env python
"""Per-kernel NODE cost of a serial (one-stream) step from a rocprofv3 kernel-trace timeline (tools/timeline.py output): the
average duration of each kernel AND the average gap to the next dispatch of the same queue -- a latency-bound chain pays
duration + gap per node, and the tracer's `--stats` table shows only the first.
    python tools/node_costs.py timeline.csv [--steps N]"""
import argparse
import collections
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps", type=int, default=1, help="steps the timeline covers (per-step columns are divided by it)")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    rows = [(float(r["start_us"]), float(r["end_us"]), r["queue"], r["kernel"]) for r in csv.DictReader(open(a.csv))]
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r[2]].append(r)
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for q, rs in byq.items():
        rs.sort()
        for i, (s, e, _, n) in enumerate(rs):
            gap = max(0.0, rs[i + 1][0] - e) if i + 1 < len(rs) else 0.0
            if gap > 200.0:                      # a segment / step boundary, not a node gap
                gap = 0.0
            g = agg[n]
            g[0] += 1
            g[1] += e - s
            g[2] += gap
    tot = sum(v[1] + v[2] for v in agg.values())
    print("%-64s %8s %9s %9s %11s %6s" % ("kernel", "calls/st", "avg us", "gap us", "us/step", "share"))
    for n, (c, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:a.top]:
        print("%-64s %8.1f %9.2f %9.2f %11.1f %5.1f%%" % (n[:64], c / a.steps, d / c, g / c, (d + g) / a.steps, 100 * (d + g) / tot))
    print("total %.1f us per step (duration + node gaps) over %d queue(s)" % (tot / a.steps, len(byq)))


if __name__ == "__main__":
    main()
