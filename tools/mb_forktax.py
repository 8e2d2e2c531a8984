#!/usr/bin/env python
"""What does a fork cost inside a hipGraph, and do two single-stream graphs on two streams overlap better?

  A  chain(N) alone, one single-stream graph
  B  chain(N) + ONE tiny kernel on a forked side stream (join at the end)         -> per-node tax of a multi-stream graph
  C  chain(N) || big(M) as two branches of one graph                               (the engine's first overlap version)
  D  chain(N) and big(M) as TWO single-stream graphs replayed on two streams        (segment scheme)
  E  big(M) alone
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd._lib import ACT_RELU, CONV_COEXIST  # noqa: E402

dev, bf = "cuda", torch.bfloat16
N = int(os.environ.get("MB_CHAIN", "1200"))
M = int(os.environ.get("MB_BIG", "24"))
w = (torch.randn(9, 64, 64, device=dev) * 0.05).to(bf)
b = torch.zeros(64, device=dev)
xa, xb = torch.randn(4, 32, 32, 64, device=dev).to(bf), torch.empty(4, 32, 32, 64, device=dev, dtype=bf)
dc = K.conv_desc(4, 32, 32, 64, 32, 32, 64, 3, 3, 1, 1, 1, 0, 1, 1, ACT_RELU)
big_in = torch.randn(76, 32, 32, 256, device=dev).to(bf)
big_out = torch.empty_like(big_in)
wb = (torch.randn(9, 256, 256, device=dev) * 0.02).to(bf)
bb = torch.zeros(256, device=dev)
db = K.conv_desc(76, 32, 32, 256, 32, 32, 256, 3, 3, 1, 1, 1, 0, 1, 1, ACT_RELU, flags=CONV_COEXIST)
tiny = torch.zeros(64, device=dev)
side = torch.cuda.Stream()


def chain():
    a, c = xa, xb
    for _ in range(N):
        K.conv_forward(dc, a, w, b, None, None, c)
        a, c = c, a


def big():
    for _ in range(M):
        K.conv_forward(db, big_in, wb, bb, None, None, big_out)


def fork(fn_side):
    def f():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            fn_side()
        chain()
        main.wait_stream(side)
    return f


def capture(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    return g


def timeit(run, reps=5):
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


gA, gE = capture(chain), capture(big)
gB = capture(fork(lambda: K.affine(tiny, tiny, 1.0, 0.0)))
gC = capture(fork(big))
tA, tE = timeit(gA.replay), timeit(gE.replay)
tB, tC = timeit(gB.replay), timeit(gC.replay)


def two_graphs():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        gE.replay()
    gA.replay()
    main.wait_stream(side)


tD = timeit(two_graphs)
hi = torch.cuda.Stream(priority=-1)           # high-priority stream for the latency-bound chain


def two_graphs_prio():
    main = torch.cuda.current_stream()
    hi.wait_stream(main)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        gE.replay()
    with torch.cuda.stream(hi):
        gA.replay()
    main.wait_stream(side)
    main.wait_stream(hi)


tP = timeit(two_graphs_prio)
print("D' two graphs, chain on a high-priority stream %.3f ms" % tP)
print("A chain(%d) %.3f ms (%.2f us/node) | B +1 forked tiny kernel %.3f ms (%.2f us/node) | E big(%d) %.3f ms | "
      "C forked graph %.3f ms | D two graphs %.3f ms | sum %.3f max %.3f" %
      (N, tA, tA * 1e3 / N, tB, tB * 1e3 / N, M, tE, tC, tD, tA + tE, max(tA, tE)))
