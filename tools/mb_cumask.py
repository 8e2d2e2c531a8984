#!/usr/bin/env python
"""CU partitioning experiment: does giving the latency-bound chain its own CUs (hipExtStreamCreateWithCUMask) beat
sharing every CU with the throughput work?  Two single-stream graphs as in the segment engine: `chain` (dependent
[4,32,32,64] 3x3 convs) and `big` (VGG-sized layers launched with TG_CONV_COEXIST)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd._lib import ACT_RELU  # noqa: E402

dev = "cuda"
bf = torch.bfloat16
NCH = int(os.environ.get("MB_CHAIN", "500"))
hip = C.CDLL("libamdhip64.so")


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) enabled."""
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, "hipExtStreamCreateWithCUMask rc=%d" % rc
    return torch.cuda.ExternalStream(st.value)


w = (torch.randn(9, 64, 64, device=dev) * 0.05).to(bf)
b = torch.zeros(64, device=dev)
xa, xb = torch.randn(4, 32, 32, 64, device=dev).to(bf), torch.empty(4, 32, 32, 64, device=dev, dtype=bf)
dc = K.conv_desc(4, 32, 32, 64, 32, 32, 64, 3, 3, 1, 1, 1, 0, 1, 1, ACT_RELU)


def layer(N, H, Cin, Cout):
    x = torch.randn(N, H, H, Cin, device=dev).to(bf)
    wt = (torch.randn(9, Cout, Cin, device=dev) * 0.05).to(bf)
    o = torch.empty(N, H, H, Cout, device=dev, dtype=bf)
    d = K.conv_desc(N, H, H, Cin, H, H, Cout, 3, 3, 1, 1, 1, 0, 1, 1, ACT_RELU, flags=K.CONV_COEXIST)
    bb = torch.zeros(Cout, device=dev)
    return lambda: K.conv_forward(d, x, wt, bb, None, None, o)


LAYERS = [layer(40, 128, 64, 64), layer(40, 64, 64, 128), layer(40, 64, 128, 128), layer(40, 32, 128, 256)] + \
         [layer(40, 32, 256, 256)] * 3 + [layer(40, 16, 256, 512)] + [layer(40, 16, 512, 512)] * 3 + [layer(40, 8, 512, 512)] * 4


def chain():
    a, c = xa, xb
    for _ in range(NCH):
        K.conv_forward(dc, a, w, b, None, None, c)
        a, c = c, a


def big():
    for _ in range(2):
        for f in LAYERS:
            f()


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
    return g


gC, gB = capture(chain), capture(big)


def run(sa, sb, reps=5):
    """chain on sa; big on sb (None = do not run it); returns ms per rep (both finished)."""
    torch.cuda.synchronize()
    e0, e1, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
    for it in range(reps + 1):
        if it == 1:
            torch.cuda.synchronize()
            e0.record(sa)
        if sb is not None and sb is not sa:
            sb.wait_stream(sa)
            with torch.cuda.stream(sb):
                gB.replay()
                eb.record(sb)
        with torch.cuda.stream(sa):
            gC.replay()
            if sb is sa:
                gB.replay()
        if sb is not None and sb is not sa:
            sa.wait_event(eb)
    e1.record(sa)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def only_big(sb, reps=5):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sb):
        gB.replay()
        e0.record(sb)
        for _ in range(reps):
            gB.replay()
        e1.record(sb)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


A, B = torch.cuda.Stream(), torch.cuda.Stream()
tc, tb = run(A, None), only_big(B)
print("unmasked: chain(%d) %.3f ms (%.2f us/node)   big %.3f ms   serial %.3f   co-run %.3f" % (NCH, tc, tc * 1e3 / NCH, tb, run(A, A), run(A, B)))
for style in ("low", "strided"):
    for n in (32, 64, 96, 128):
        if style == "low":
            cbits = list(range(n))
        else:
            step = 256 // n
            cbits = list(range(0, 256, step))[:n]
        rest = [i for i in range(256) if i not in set(cbits)]
        try:
            sa, sb = masked_stream(cbits), masked_stream(rest)
            t1, t2 = run(sa, None), only_big(sb)
            print("%-8s chain on %3d CUs: chain alone %.3f ms (%.2f us/node)  big on %3d CUs alone %.3f ms   co-run %.3f ms" %
                  (style, n, t1, t1 * 1e3 / NCH, len(rest), t2, run(sa, sb)))
        except Exception as e:                # noqa: BLE001
            print("%-8s %d: failed: %s" % (style, n, e))
