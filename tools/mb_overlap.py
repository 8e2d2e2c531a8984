#!/usr/bin/env python
"""Can the latency-bound recurrent chain share the chip with throughput work?  (overlap experiment)

chain  = N dependent 3x3 64->64 convs at [4,32,32,64] (the generator's res-block chain: <4,16> tiles, 36 KB LDS,
         ~120 registers, one wave per SIMD, ~3.4 us per node of which ~1.7 us is the graph-node floor);
big    = M independent-of-the-chain 3x3 64->64 convs at [76,128,128,64] (VGG-19 conv1_2 shape).
Three hipGraphs: chain alone, big alone, both as parallel branches.  If the branches overlap, t(both) ~ max, else ~ sum.
Run with TG_C3_MAXTH=16 (big = <16,64>: 130 KB LDS, ~400 registers -> no chain workgroup fits beside it) and
TG_C3_MAXTH=8 (<8,64>: 109 KB, ~300 registers -> a chain workgroup fits)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd._lib import ACT_RELU  # noqa: E402

dev = "cuda"
bf = torch.bfloat16
NCH = int(os.environ.get("MB_CHAIN", "400"))
NBIG = int(os.environ.get("MB_BIG", "12"))
w = (torch.randn(9, 64, 64, device=dev) * 0.05).to(bf)
b = torch.zeros(64, device=dev)
xa, xb = torch.randn(4, 32, 32, 64, device=dev).to(bf), torch.empty(4, 32, 32, 64, device=dev, dtype=bf)
dc = K.conv_desc(4, 32, 32, 64, 32, 32, 64, 3, 3, 1, 1, 1, 0, 1, 1, ACT_RELU)
big_in = torch.randn(76, 128, 128, 64, device=dev).to(bf)
big_out = torch.empty_like(big_in)
db = K.conv_desc(76, 128, 128, 64, 128, 128, 64, 3, 3, 1, 1, 1, 0, 1, 1, ACT_RELU)


def chain():
    a, c = xa, xb
    for _ in range(NCH):
        K.conv_forward(dc, a, w, b, None, None, c)
        a, c = c, a


def big():
    for _ in range(NBIG):
        K.conv_forward(db, big_in, w, b, None, None, big_out)


side = torch.cuda.Stream()


def both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        big()
    chain()
    main.wait_stream(side)


def graph_time(fn, reps=5):
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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tc, tb, tboth = graph_time(chain), graph_time(big), graph_time(both)
print("TG_C3_MAXTH=%s  chain(%d) %.3f ms (%.2f us/node)  big(%d) %.3f ms (%.1f us each)  both %.3f ms  sum %.3f  max %.3f  hidden %.0f%%" %
      (os.environ.get("TG_C3_MAXTH", "16"), NCH, tc, tc * 1e3 / NCH, NBIG, tb, tb * 1e3 / NBIG, tboth, tc + tb, max(tc, tb),
       100.0 * (tc + tb - tboth) / min(tc, tb)))
