#!/usr/bin/env python
"""Per-launch timings (HIP events on the launch stream) of the hot kernels at the shapes of the BASELINE configs."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd._lib import ACT_RELU  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def graph_timeit(fn, chain=50, iters=20):
    """The same launch `chain` times inside one hipGraph: per-launch time without host launch overhead."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(chain):
            fn()
    return timeit(g.replay, iters, 3) / chain


def conv_case(N, H, W, Cin, Cout, k=3, s=1, mode=0, dtype=torch.bfloat16, out_hw=None):
    x = torch.randn(N, H, W, Cin, device=DEV).to(dtype)
    w = (torch.randn(k * k, Cout, Cin, device=DEV) * 0.05).to(dtype)
    b = torch.zeros(Cout, device=DEV)
    if out_hw is None:
        Ho, pt = K.same_pad(H, k, s)
        Wo, pl = K.same_pad(W, k, s)
    else:
        (Ho, Wo), pt, pl = out_hw, 0, 0
    out = torch.empty(N, Ho, Wo, Cout, device=DEV, dtype=dtype)
    d = K.conv_desc(N, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, mode, K.dt(x), K.dt(out), ACT_RELU,
                    flags=int(os.environ.get("MB_CONV_FLAGS", "0")))      # 1 TG_CONV_COEXIST
    flops = 2.0 * N * Ho * Wo * Cout * k * k * Cin if mode == 0 else 2.0 * N * H * W * Cout * k * k * Cin
    return (lambda: K.conv_forward(d, x, w, b, None, None, out)), flops


def wgrad_case(N, H, W, Cin, Cout, k=3, s=1, dtype=torch.bfloat16):
    x = torch.randn(N, H, W, Cin, device=DEV).to(dtype)
    Ho, pt = K.same_pad(H, k, s)
    Wo, pl = K.same_pad(W, k, s)
    gy = torch.randn(N, Ho, Wo, Cout, device=DEV).to(dtype)
    dw = torch.zeros(k, k, Cin, Cout, device=DEV)
    db = torch.zeros(Cout, device=DEV)
    d = K.conv_desc(N, H, W, Cin, Ho, Wo, Cout, k, k, s, pt, pl, 0, 0, 0)
    return (lambda: K.conv_wgrad(d, x, gy, dw, db)), 2.0 * N * Ho * Wo * Cout * k * k * Cin


def graph_chain_probes():
    """--graph: three questions about dependent kernel chains inside a hipGraph (rounds 2-3, DESIGN lessons 1, 5): the per-node
    floor, whether it grows with the graph's length, whether alternating kernels cost more than homogeneous chains."""
    from tecogan_amd import kernels as K
    a, b = torch.ones(64, device="cuda"), torch.ones(64, device="cuda")
    print("graph node floor (tiny dependent kernel): %.2f us" % graph_timeit(lambda: K.lincomb(a, b, a, 0.5, 0.5)))
    n = 4 * 32 * 32 * 64
    x, y = torch.ones(n, device="cuda"), torch.ones(n, device="cuda")
    print("1 MiB fp32 lincomb node: %.2f us" % graph_timeit(lambda: K.lincomb(x, y, x, 0.5, 0.5)))
    fn, _ = conv_case(4, 32, 32, 64, 64)
    for chain in (50, 200, 600, 1500, 3000):
        print("chain %5d nodes: %.3f us per conv node" % (chain, graph_timeit(fn, chain=chain, iters=10)))
    c8, _ = conv_case(4, 128, 128, 8, 64)
    wg, _ = wgrad_case(40, 32, 32, 64, 64)
    ta, tb, tc = (graph_timeit(f, chain=100, iters=10) for f in (fn, c8, wg))
    print("alone: LR conv %.2f  c8 conv %.2f  wgrad %.2f us" % (ta, tb, tc))
    for name, f, g, s in (("LRconv+c8", fn, c8, ta + tb), ("LRconv+wgrad", fn, wg, ta + tc)):
        def pair(f=f, g=g):
            f()
            g()
        print("%-14s alternating %.2f us per pair vs %.2f summed" % (name, graph_timeit(pair, chain=100, iters=10), s))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="", help="substring filter on the case name")
    ap.add_argument("--graph", action="store_true", help="the hipGraph chain probes instead of the kernel table")
    a = ap.parse_args()
    if a.graph:
        return graph_chain_probes()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    cases = [
        ("conv3x3 gen  [4,32,32,64->64]", conv_case(4, 32, 32, 64, 64, dtype=dt)),
        ("conv3x3 gen-in [4,32,32,56->64]", conv_case(4, 32, 32, 56, 64, dtype=dt)),
        ("conv3x3 out  [4,128,128,64->3]", conv_case(4, 128, 128, 64, 3, dtype=dt)),
        ("deconv fwd   [4,32,32,64]->64x64", conv_case(4, 32, 32, 64, 64, 3, 2, 1, dt, (64, 64))),
        ("deconv fwd   [4,64,64,64]->128x128", conv_case(4, 64, 64, 64, 64, 3, 2, 1, dt, (128, 128))),
        ("deconv bwd   [4,128,128,64]->64x64 (gather s2)", conv_case(4, 128, 128, 64, 64, 3, 2, 0, dt, (64, 64))),
        ("conv3x3 fnet [36,32,32,32->32]", conv_case(36, 32, 32, 32, 32, dtype=dt)),
        ("conv3x3 fnet [36,4,4,256->256]", conv_case(36, 4, 4, 256, 256, dtype=dt)),
        ("conv3x3 vgg  [76,128,128,64->64]", conv_case(76, 128, 128, 64, 64, dtype=dt)),
        ("conv3x3 vgg  [76,32,32,256->256]", conv_case(76, 32, 32, 256, 256, dtype=dt)),
        ("conv3x3 vgg  [76,16,16,512->512]", conv_case(76, 16, 16, 512, 512, dtype=dt)),
        ("conv3x3 vgg5 [76,8,8,512->512]", conv_case(76, 8, 8, 512, 512, dtype=dt)),
        ("conv3x3 fnet [72,8,8,128->128]", conv_case(72, 8, 8, 128, 128, dtype=dt)),
        ("conv3x3 vgg5 [32,8,8,512->512]", conv_case(32, 8, 8, 512, 512, dtype=dt)),
        ("conv3x3 vgg5 [44,8,8,512->512]", conv_case(44, 8, 8, 512, 512, dtype=dt)),
        ("conv3x3 fnet [72,4,4,256->256]", conv_case(72, 4, 4, 256, 256, dtype=dt)),
        ("conv4x4s2 D  [24,128,128,64->64]", conv_case(24, 128, 128, 64, 64, 4, 2, 0, dt)),
        ("conv3x3 inf  [1,270,480,64->64]", conv_case(1, 270, 480, 64, 64, dtype=dt)),
        ("conv3x3 inf  [1,270,480,56->64]", conv_case(1, 270, 480, 56, 64, dtype=dt)),
        ("conv3x3 D-in [24,128,128,32->64]", conv_case(24, 128, 128, 32, 64, dtype=dt)),
        ("conv3x3 fnet [72,32,32,64->64]", conv_case(72, 32, 32, 64, 64, dtype=dt)),
        ("conv3x3 inf  [1,1080,1920,64->3]", conv_case(1, 1080, 1920, 64, 3, dtype=dt)),
        ("conv3x3 wide [40,64,64,128->128]", conv_case(40, 64, 64, 128, 128, dtype=dt)),
        ("conv3x3 wide [20,64,64,128->128]", conv_case(20, 64, 64, 128, 128, dtype=dt)),
        ("conv3x3 wide [40,32,32,256->256]", conv_case(40, 32, 32, 256, 256, dtype=dt)),
        ("conv3x3 wide [20,32,32,256->256]", conv_case(20, 32, 32, 256, 256, dtype=dt)),
        ("conv3x3 wide [40,16,16,512->512]", conv_case(40, 16, 16, 512, 512, dtype=dt)),
        ("conv3x3 wide [20,16,16,512->512]", conv_case(20, 16, 16, 512, 512, dtype=dt)),
        ("conv3x3 wide [72,16,16,128->128]", conv_case(72, 16, 16, 128, 128, dtype=dt)),
        ("conv3x3 wide [76,64,64,64->128]", conv_case(76, 64, 64, 64, 128, dtype=dt)),
        ("wgrad gen    [40,32,32,64->64]", wgrad_case(40, 32, 32, 64, 64, dtype=dt)),
        ("wgrad tran2  [40,128,128,64] s2", wgrad_case(40, 128, 128, 64, 64, 3, 2, dtype=dt)),
        ("wgrad fnet   [36,32,32,32->32]", wgrad_case(36, 32, 32, 32, 32, dtype=dt)),
        ("wgrad fnet   [36,4,4,256->256]", wgrad_case(36, 4, 4, 256, 256, dtype=dt)),
        ("wgrad D      [24,128,128,64->64] k4s2", wgrad_case(24, 128, 128, 64, 64, 4, 2, dtype=dt)),
    ]
    print("%-52s %10s %10s %10s" % ("case (%s)" % a.dtype, "eager us", "graph us", "TFLOP/s"))
    for name, (fn, flops) in cases:
        if a.only and a.only not in name:
            continue
        t1 = timeit(fn)
        t2 = graph_timeit(fn)
        print("%-52s %10.2f %10.2f %10.1f" % (name, t1, t2, flops / t2 / 1e6))


if __name__ == "__main__":
    main()
