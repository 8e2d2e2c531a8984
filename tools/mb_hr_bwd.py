#!/usr/bin/env python
"""The BPTT's HR tail at the training shape (t1 = [4,64,64,64], HR 128x128): one launch (csrc/hr_bwd_lat.hip) against the three
launches it replaces (tg_concat2_pad, the 8-channel input-gradient conv, the gather-form transposed-conv gradient); graph-chained."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd._lib import ACT_RELU, TG_BF16  # noqa: E402
from tools.microbench import graph_timeit  # noqa: E402

DEV = "cuda"
torch.manual_seed(0)
for N, H2, W2 in ((4, 64, 64), (1, 64, 64), (8, 64, 64)):
    Ho, Wo = 2 * H2, 2 * W2
    bf = torch.bfloat16
    d_frame = torch.randn(N, Ho, Wo, 3, device=DEV) * 0.01
    wo = (torch.randn(9, 64, 8, device=DEV) * 0.05).to(bf)
    wtr = (torch.randn(9, 64, 64, device=DEV) * 0.05).to(bf)
    wtr_f = K.frag_order(wtr)
    t2, t1 = torch.randn(N, Ho, Wo, 64, device=DEV).to(bf), torch.randn(N, H2, W2, 64, device=DEV).to(bf)
    g_out = torch.empty(N, Ho, Wo, 8, device=DEV, dtype=bf)
    g_t2, g_t1 = torch.empty_like(t2), torch.empty_like(t1)
    dA = K.conv_desc(N, Ho, Wo, 8, Ho, Wo, 64, 3, 3, 1, 1, 1, 1, TG_BF16, TG_BF16, 0, 0.0, ACT_RELU, 0.0)
    dB = K.conv_desc(N, Ho, Wo, 64, H2, W2, 64, 3, 3, 2, 0, 0, 0, TG_BF16, TG_BF16, 0, 0.0, ACT_RELU, 0.0)

    def three():
        K.concat2_pad(d_frame, None, g_out, scale=2.0)
        K.conv_forward(dA, g_out, wo, None, None, t2, g_t2)
        K.conv_forward(dB, g_t2, wtr, None, None, t1, g_t1)

    def one():
        K.hr_tail_backward(d_frame, 2.0, wo, t2, wtr_f, t1, g_out, g_t2, g_t1)

    print("HR tail input gradients, t1 [%d,%d,%d,64]: three launches %6.2f us   one launch %6.2f us (graph chain of 20)"
          % (N, H2, W2, graph_timeit(three, chain=20), graph_timeit(one, chain=20)), flush=True)
