#!/usr/bin/env python
"""The fused warp kernels alone (graph-chained launches): forward at the 1080p inference shape and the training shape,
backward (scatter, with the neighbour hand-over) at the training shape with smooth / FNet-like / random flows."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tools.microbench import graph_timeit  # noqa: E402

DEV = "cuda"
torch.manual_seed(0)
for B, h, w in ((1, 270, 480), (4, 32, 32)):
    pre = torch.rand(B, 4 * h, 4 * w, 3, device=DEV)
    lr = torch.rand(B, h, w, 3, device=DEV)
    for kind, flow in (("smooth", torch.full((B, h, w, 2), 0.31, device=DEV) + 0.02 * torch.randn(B, h, w, 2, device=DEV)),
                       ("random3", 3.0 * torch.randn(B, h, w, 2, device=DEV))):
        out = torch.empty(B, h, w, 56, device=DEV, dtype=torch.bfloat16)
        t = graph_timeit(lambda: K.warp_s2d_forward(pre, flow, lr, out, 0.5, 0.5), chain=20)
        by = B * h * w * (16 * 12 + 8 + 12 + 56 * 2)
        print("warp_s2d_fwd [%d,%d,%d] %-8s %7.2f us  %6.2f TB/s algorithmic" % (B, h, w, kind, t, by / t / 1e6), flush=True)
B, h, w = 4, 32, 32
pre = torch.rand(B, 4 * h, 4 * w, 3, device=DEV)
g = torch.randn(B, h, w, 56, device=DEV).bfloat16()
for kind, flow in (("smooth", torch.full((B, h, w, 2), 0.31, device=DEV) + 0.02 * torch.randn(B, h, w, 2, device=DEV)),
                   ("fnet-like", 1.5 * torch.randn(B, h, w, 2, device=DEV)),
                   ("random12", 12.0 * torch.randn(B, h, w, 2, device=DEV))):
    d_pre = torch.zeros(B, 4 * h, 4 * w, 3, device=DEV)
    d_flow = torch.zeros(B, h, w, 2, device=DEV)
    t = graph_timeit(lambda: K.warp_s2d_backward(g, pre, flow, d_pre, d_flow, 0.5), chain=20)
    print("warp_s2d_bwd [%d,%d,%d] %-9s merge=%s %7.2f us" % (B, h, w, kind, "1", t), flush=True)
