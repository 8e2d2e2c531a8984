#!/usr/bin/env python
"""The wide VGG-19 layers (conv2_2 ... conv4_4 and their input gradients) at the image counts of the TecoGAN step's three VGG
passes (28 / 40 / 48 / 76 images of 128x128), graph-timed per launch: tg_conv_forward (conv3x3_dma.hip: both operands by LDS-DMA)
against tg_conv3x3_wide_frag (conv3x3_wr.hip: weights global -> registers) with 16- and 8-row tiles.
    python tools/mb_wide.py [--n 28 48] [--only 256]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd._lib import ACT_NONE, ACT_RELU, TG_BF16  # noqa: E402
from tools.microbench import graph_timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs="*", default=[28, 40, 48, 76])
ap.add_argument("--only", default="")
ap.add_argument("--chain", type=int, default=20)
ap.add_argument("--c5", action="store_true", help="VGG conv5_x (8x8 images: packed tiles) with K splits 1 / 2 / 4 / auto")
ap.add_argument("--c64", action="store_true", help="the 64-channel layers instead (conv3x3_ws.hip's: VGG conv1_2 / conv2_1, the 1080p res-block conv)")
a = ap.parse_args()
DEV = "cuda"
LAYERS = [  # name, HW, Cin, Cout, input-gradient form (mask)
    ("conv2_2      ", 64, 128, 128, False), ("conv3_1      ", 32, 128, 256, False), ("conv3_x      ", 32, 256, 256, False),
    ("conv4_1      ", 16, 256, 512, False), ("conv4_x      ", 16, 512, 512, False),
    ("conv2_1 dX   ", 64, 128, 64, True), ("conv2_2 dX   ", 64, 128, 128, True), ("conv3_1 dX   ", 32, 256, 128, True),
    ("conv3_x dX   ", 32, 256, 256, True), ("conv4_1 dX   ", 16, 512, 256, True), ("conv4_x dX   ", 16, 512, 512, True),
]
if a.c64:
    LAYERS = [("conv1_2      ", 128, 64, 64, False), ("conv2_1      ", 64, 64, 128, False), ("conv1_2 dX   ", 128, 64, 64, True),
              ("1080p block  ", (270, 480), 64, 64, False)]
if a.c5:
    LAYERS = [("conv5_x      ", 8, 512, 512, False), ("conv5_x dX   ", 8, 512, 512, True)]
tot = {}
for N in a.n:
    for name, HW, Cin, Cout, bwd in LAYERS:
        if isinstance(HW, tuple):
            (HW, WW), N = HW, 1
        else:
            WW = HW
        if a.only and a.only not in name + " %d %d" % (Cin, Cout):
            continue
        x = torch.randn(N, HW, WW, Cin, device=DEV).bfloat16()
        w = (torch.randn(9, Cout, Cin, device=DEV) * 0.05).bfloat16()
        wf = K.pack_wide_frag(w, torch.empty_like(w), Cout, Cin, bwd)
        b = None if bwd else torch.zeros(Cout, device=DEV)
        aux = torch.randn(N, HW, WW, Cout, device=DEV).bfloat16() if bwd else None
        out = torch.empty(N, HW, WW, Cout, device=DEV, dtype=torch.bfloat16)
        d = K.conv_desc(N, HW, WW, Cin, HW, WW, Cout, 3, 3, 1, 1, 1, 1 if bwd else 0, TG_BF16, TG_BF16,
                        ACT_NONE if bwd else ACT_RELU, 0.0, ACT_RELU if bwd else ACT_NONE, 0.0)
        fl = 2.0 * N * HW * WW * Cout * 9 * Cin
        t_old = graph_timeit(lambda: K.conv_forward(d, x, w, b, None, aux, out), a.chain, 10)
        if HW == 8:
            ts = [graph_timeit(lambda: K.conv3x3_wide_frag(d, x, wf, b, None, aux, out, 0, ks), a.chain, 10) for ks in (1, 2, 4, 0)]
            print("N=%2d %s [%3d,%3d->%3d]  dma %6.1f us %5.0f TF/s | ks1 %6.1f us | ks2 %6.1f | ks4 %6.1f | auto %6.1f us %5.0f TF/s"
                  % (N, name, HW, Cin, Cout, t_old, fl / t_old * 1e-6, ts[0], ts[1], ts[2], ts[3], fl / ts[3] * 1e-6), flush=True)
            continue
        t16 = graph_timeit(lambda: K.conv3x3_wide_frag(d, x, wf, b, None, aux, out, 16), a.chain, 10)
        t8 = graph_timeit(lambda: K.conv3x3_wide_frag(d, x, wf, b, None, aux, out, 8), a.chain, 10)
        t0 = graph_timeit(lambda: K.conv3x3_wide_frag(d, x, wf, b, None, aux, out, 0), a.chain, 10)
        for k, t in (("dma", t_old), ("wr16", t16), ("wr8", t8), ("auto", t0)):
            tot[(N, k)] = tot.get((N, k), 0.0) + t * (3 if "_x" in name else 1)
        print("N=%2d %s [%3d,%3d->%3d]  dma %6.1f us %5.0f TF/s | wr16 %6.1f us %5.0f | wr8 %6.1f us %5.0f | auto %6.1f"
              % (N, name, HW, Cin, Cout, t_old, fl / t_old * 1e-6, t16, fl / t16 * 1e-6, t8, fl / t8 * 1e-6, t0), flush=True)
for N in a.n:
    print("N=%2d wide layers of one pass (conv3_x / conv4_x three times): dma %7.1f us | wr16 %7.1f | wr8 %7.1f | auto %7.1f"
          % (N, tot.get((N, "dma"), 0), tot.get((N, "wr16"), 0), tot.get((N, "wr8"), 0), tot.get((N, "auto"), 0)))
