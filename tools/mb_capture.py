#!/usr/bin/env python
"""Which cross-stream patterns survive hipGraph stream capture on this stack?  (The overlap schedule of engine.py forks
a side stream several times and hands events both ways; capture_end crashed on the first version.)  Each pattern runs in
its own process; a crash of one does not stop the others.

    python tools/mb_capture.py            # run all patterns
    python tools/mb_capture.py P3         # run one (child mode)
"""
import subprocess
import sys

PATTERNS = ["P0", "P1", "P2", "P3", "P4", "P5", "P6", "P7", "P8", "P9"]


def child(name):
    import torch
    dev = "cuda"
    a = torch.zeros(1 << 20, device=dev)
    b = torch.zeros(1 << 20, device=dev)
    c = torch.zeros(1 << 20, device=dev)
    side = torch.cuda.Stream()

    def k(t):
        t.add_(1.0)

    def ev(stream):
        e = torch.cuda.Event()
        e.record(stream)
        return e

    keep = []

    def prog():
        main = torch.cuda.current_stream()
        if name == "P0":                       # one fork, one join
            side.wait_stream(main)
            with torch.cuda.stream(side):
                k(a)
            k(b)
            main.wait_stream(side)
        elif name == "P1":                     # event side->main, then MORE side work, then join
            side.wait_stream(main)
            with torch.cuda.stream(side):
                k(a)
                e = ev(side)
                k(c)
            k(b)
            main.wait_event(e)
            k(a)
            main.wait_stream(side)
            keep.append(e)
        elif name == "P2":                     # event side->main, no further side work, join
            side.wait_stream(main)
            with torch.cuda.stream(side):
                k(a)
                e = ev(side)
            k(b)
            main.wait_event(e)
            k(a)
            main.wait_stream(side)
            keep.append(e)
        elif name == "P3":                     # two forks (main->side edges), one join
            side.wait_stream(main)
            with torch.cuda.stream(side):
                k(a)
            k(b)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                k(c)
            k(b)
            main.wait_stream(side)
        elif name == "P4":                     # P1 + another fork after main consumed the event
            side.wait_stream(main)
            with torch.cuda.stream(side):
                k(a)
                e = ev(side)
                k(c)
            k(b)
            main.wait_event(e)
            k(a)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                k(c)
            k(b)
            main.wait_stream(side)
            keep.append(e)
        elif name == "P5":                     # a stream waits on its own event
            side.wait_stream(main)
            with torch.cuda.stream(side):
                k(a)
                e = ev(side)
                side.wait_event(e)
                k(a)
            k(b)
            main.wait_stream(side)
            keep.append(e)
        elif name == "P6":                     # two side->main events consumed at different times
            side.wait_stream(main)
            with torch.cuda.stream(side):
                k(a)
                e1 = ev(side)
                k(c)
                e2 = ev(side)
                k(c)
            k(b)
            main.wait_event(e1)
            k(a)
            main.wait_event(e2)
            k(c)
            main.wait_stream(side)
            keep.extend([e1, e2])
        elif name == "P7":                     # allocation on the side stream inside the capture, used on main
            side.wait_stream(main)
            with torch.cuda.stream(side):
                t = torch.empty(1 << 20, device=dev)
                t.fill_(2.0)
                e = ev(side)
            main.wait_event(e)
            a.add_(t)
            main.wait_stream(side)
            keep.extend([t, e])
        elif name == "P8":                     # event side->main, then a fork, side work that outlives main's last node
            side.wait_stream(main)
            with torch.cuda.stream(side):
                k(a)
                e = ev(side)
            main.wait_event(e)
            k(b)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(8):
                    k(c)
            main.wait_stream(side)
            keep.append(e)
        elif name == "P9":                     # the engine's shape: 3 forks, 3 side->main events, interleaved
            side.wait_stream(main)
            with torch.cuda.stream(side):
                k(a)
                e1 = ev(side)
            k(b)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                k(c)
                e2 = ev(side)
            k(b)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                side.wait_event(e1)
                k(a)
                e3 = ev(side)
            k(b)
            main.wait_event(e2)
            k(c)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                k(c)
            main.wait_event(e1)
            main.wait_event(e3)
            k(a)
            main.wait_stream(side)
            keep.extend([e1, e2, e3])

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        prog()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        prog()
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    print("%s ok a=%g b=%g c=%g" % (name, a[0].item(), b[0].item(), c[0].item()))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for p in PATTERNS:
            r = subprocess.run([sys.executable, __file__, p], capture_output=True, text=True, timeout=120)
            tail = (r.stdout.strip().splitlines() or [""])[-1]
            print("%s rc=%d %s" % (p, r.returncode, tail if r.returncode == 0 else (r.stderr.strip().splitlines() or ["?"])[0][:120]), flush=True)
