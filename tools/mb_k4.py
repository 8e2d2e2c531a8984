#!/usr/bin/env python
"""The discriminator's 4x4 stride-2 convs (lib/Teco.py:52-66) and their input gradients at the TecoGAN step's shapes (24 triplets
per pass, 48 in the merged own-gradient pass), graph-timed per launch: tg_conv_forward (conv_igemm.hip) against tg_conv4x4s2_frag.
    python tools/mb_k4.py [--n 24 48]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd._lib import ACT_NONE, TG_BF16  # noqa: E402
from tools.microbench import graph_timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs="*", default=[24, 48])
ap.add_argument("--chain", type=int, default=20)
a = ap.parse_args()
DEV = "cuda"
LAYERS = [("disblock_1", 128, 64, 64), ("disblock_3", 64, 64, 64), ("disblock_5", 32, 64, 128), ("disblock_7", 16, 128, 256)]
for N in a.n:
    for name, HW, Cin, Cout in LAYERS:
        Ho = HW // 2
        x = torch.randn(N, HW, HW, Cin, device=DEV).bfloat16()
        gy = torch.randn(N, Ho, Ho, Cout, device=DEV).bfloat16()
        wt = (torch.randn(16, Cout, Cin, device=DEV) * 0.05).bfloat16()
        wn = (torch.randn(16, Cin, Cout, device=DEV) * 0.05).bfloat16()
        wtf = K.pack_taps_frag(wt, torch.empty_like(wt), 16, Cout, Cin)
        wnf = K.pack_taps_frag(wn, torch.empty_like(wn), 16, Cin, Cout)
        out = torch.empty(N, Ho, Ho, Cout, device=DEV, dtype=torch.bfloat16)
        dx = torch.empty(N, HW, HW, Cin, device=DEV, dtype=torch.bfloat16)
        df = K.conv_desc(N, HW, HW, Cin, Ho, Ho, Cout, 4, 4, 2, 1, 1, 0, TG_BF16, TG_BF16, ACT_NONE)
        db = K.conv_desc(N, Ho, Ho, Cout, HW, HW, Cin, 4, 4, 2, 1, 1, 1, TG_BF16, TG_BF16, ACT_NONE)
        fl = 2.0 * N * Ho * Ho * Cout * 16 * Cin
        by = (x.numel() + out.numel()) * 2.0
        t = [graph_timeit(f, a.chain, 10) for f in (
            lambda: K.conv_forward(df, x, wt, None, None, None, out), lambda: K.conv4x4s2_frag(df, x, wtf, None, None, None, out),
            lambda: K.conv_forward(db, gy, wn, None, None, None, dx), lambda: K.conv4x4s2_frag(db, gy, wnf, None, None, None, dx))]
        print("N=%2d %s [%3d,%3d->%3d] fwd: igemm %6.1f us -> %6.1f us (%5.0f TF/s, %4.2f TB/s) | dX: igemm %6.1f us -> %6.1f us (%5.0f TF/s, %4.2f TB/s)"
              % (N, name, HW, Cin, Cout, t[0], t[1], fl / t[1] * 1e-6, by / t[1] * 1e-6, t[2], t[3], fl / t[3] * 1e-6, by / t[3] * 1e-6), flush=True)
