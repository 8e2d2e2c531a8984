#!/usr/bin/env python
"""One VGG-19 perceptual-loss pass as the TecoGAN step runs it (lib/Teco.py:174-178,339-359: forward of N generated frames, cosine
loss of the four taps against target features, input gradient) for N images of 128x128, graph-timed; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel table of a pass.
    python tools/mb_vgg.py [--n 32 44 76] [--fwd-only]"""
import argparse
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd.nets import VGG19, VGG_CPAD, VGG_TAPS  # noqa: E402
from tecogan_amd.params import ParamStore, init_values, vgg_spec  # noqa: E402
from tools.microbench import timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs="*", default=[20, 32, 44, 76])
ap.add_argument("--fwd-only", action="store_true")
ap.add_argument("--flags", type=int, default=0, help="1 = TG_CONV_COEXIST tiles")
ap.add_argument("--no-wide", action="store_true", help="wide layers on conv3x3_dma.hip instead of conv3x3_wr.hip (the round-4 path)")
a = ap.parse_args()
if a.no_wide:
    import tecogan_amd.nets as _nets
    _nets.WIDE_FRAG = False
dev = "cuda"
vps = ParamStore(OrderedDict(vgg=vgg_spec()), dev, torch.bfloat16, trainable=False, wide_frag=True)
vps.load(init_values(vgg_spec(), 45, he_normal=True))
V = VGG19(vps)
H = 128
for N in a.n:
    x = torch.rand(N, H, H, 3, device=dev) * 2 - 1
    taps_t = {VGG_TAPS[0]: (H // 2, 128), VGG_TAPS[1]: (H // 4, 256), VGG_TAPS[2]: (H // 8, 512), VGG_TAPS[3]: (H // 16, 512)}
    taps_t = {k: torch.randn(N, hw, hw, c, device=dev).bfloat16() for k, (hw, c) in taps_t.items()}
    slot = torch.zeros(4, device=dev)
    dst = torch.zeros(N, H, H, 3, device=dev)

    def one_pass():
        xg = K.vgg_preprocess_forward(x, torch.empty(N, H, H, VGG_CPAD, device=dev, dtype=torch.bfloat16))
        taps_g, acts = V.forward(xg, keep=not a.fwd_only, flags=a.flags)
        if a.fwd_only:
            return
        d_taps = {}
        for i, key in enumerate(VGG_TAPS):
            g = taps_g[key]
            d = torch.empty_like(g)
            K.cosine_loss(g, taps_t[key], 1e-3, -1e-3, slot[i:i + 1], d)
            d_taps[key] = d
        dx = V.backward(acts, d_taps, flags=a.flags)
        K.vgg_preprocess_backward(dx, dst)

    for _ in range(2):
        one_pass()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        one_pass()
    t = timeit(g.replay, 20, 3)
    gm = 6.33 * N * (1 if a.fwd_only else 2)
    print("VGG-19 pass (%s), %3d images 128x128, flags %d: %8.1f us = %5.1f us per image, %6.1f TFLOP/s"
          % ("forward" if a.fwd_only else "forward + cosine loss + input gradient", N, a.flags, t, t / N, 2 * gm / t * 1e-3), flush=True)
