#!/usr/bin/env python
"""Segment schedule of a replayed training step, read from device wall-clock stamps captured at every segment boundary
(TG_SEG_STAMPS=1; no tracing tool attached, so launch gaps are the real ones).
    python tools/seg_timeline.py [--config tecogan] [--steps 20]"""
import argparse
import os
import sys

os.environ["TG_SEG_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="tecogan")
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    eng = bench.new_engine(a.config, "bf16", dev)
    F = bench.make_flags(a.config)
    eng.set_batch(*bench.synthetic_batch(F, 1, dev))
    import time
    prev = torch.zeros_like(eng.seg_stamps)
    host = []
    for i in range(a.steps):
        if i == a.steps - 1:
            prev.copy_(eng.seg_stamps)          # on the main stream, between the two steps: the previous step's stamps
        h0 = time.perf_counter()
        eng.step(**({"next_targets": True} if getattr(eng, "lookahead", False) else {}))
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    t = eng.seg_stamps.cpu().tolist()
    tp = prev.cpu().tolist()
    names = sorted(eng.seg_stamp_names.items(), key=lambda kv: t[2 * kv[1]])
    t0 = min(t[2 * i] for _, i in names)
    skey = {s["name"]: s["skey"] for s in (eng._segs or [])}
    print("segment schedule of the last of %d replayed steps (ms from the first segment's start; 100 MHz device clock)" % a.steps)
    print("%-12s %-3s %9s %9s %9s" % ("segment", "st", "start", "end", "length"))
    for n, i in names:
        s, e = (t[2 * i] - t0) / 1e5, (t[2 * i + 1] - t0) / 1e5
        print("%-12s %-3s %9.3f %9.3f %9.3f" % (n, skey.get(n, "M"), s, e, e - s))
    print("the step before it (same clock origin):")
    for n, i in names:
        s_, e_ = (tp[2 * i] - t0) / 1e5, (tp[2 * i + 1] - t0) / 1e5
        print("%-12s %-3s %9.3f %9.3f %9.3f" % (n, skey.get(n, "M"), s_, e_, e_ - s_))
    last_prev = max(tp[2 * i + 1] for _, i in names)
    print("previous step's last segment end -> this step's first segment start: %.3f ms" % ((t0 - last_prev) / 1e5))
    print("host time inside step() (ms), last 6 steps: %s" % " ".join("%.2f" % (h * 1e3) for h in host[-6:]))
    # idle time of the main stream between consecutive M segments
    ms = [(t[2 * i], t[2 * i + 1], n) for n, i in names if skey.get(n, "M") == "M"]
    for (s0, e0, n0), (s1, e1, n1) in zip(ms, ms[1:]):
        print("main stream idle between %-8s and %-8s: %7.3f ms" % (n0, n1, (s1 - e0) / 1e5))


if __name__ == "__main__":
    main()
