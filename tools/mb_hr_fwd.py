#!/usr/bin/env python
"""The forward HR tail at the training shape: conv_tran1 [4,32,32,64] -> [4,64,64,64] and conv_tran2 + output conv + bicubic skip
[4,64,64,64] -> [4,128,128,3], latency-regime launches (csrc/hr_fwd_lat.hip) against the launches they replace; graph-chained."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd._lib import ACT_NONE, ACT_RELU, TG_BF16, TG_F32  # noqa: E402
from tools.microbench import graph_timeit  # noqa: E402

DEV = "cuda"
torch.manual_seed(0)
bf = torch.bfloat16
for N in (4, 1, 8):
    a = torch.randn(N, 32, 32, 64, device=DEV).to(bf)
    w1 = (torch.randn(9, 64, 64, device=DEV) * 0.05).to(bf)
    w2 = (torch.randn(9, 64, 64, device=DEV) * 0.05).to(bf)
    w3 = (torch.randn(9, 3, 64, device=DEV) * 0.05).to(bf)
    b = torch.zeros(64, device=DEV)
    b3 = torch.zeros(3, device=DEV)
    t1 = torch.empty(N, 64, 64, 64, device=DEV, dtype=bf)
    t2 = torch.empty(N, 128, 128, 64, device=DEV, dtype=bf)
    c = torch.empty(N, 128, 128, 3, device=DEV)
    out = torch.empty(N, 128, 128, 3, device=DEV)
    x_in = torch.randn(N, 32, 32, 56, device=DEV).to(bf)
    f1, f2 = K.frag_order(w1), K.frag_order(w2)
    d1 = K.conv_desc(N, 32, 32, 64, 64, 64, 64, 3, 3, 2, 0, 0, 1, TG_BF16, TG_BF16, ACT_RELU, 0.0, flags=1)
    d2 = K.conv_desc(N, 64, 64, 64, 128, 128, 64, 3, 3, 2, 0, 0, 1, TG_BF16, TG_BF16, ACT_RELU, 0.0, flags=1)
    d3 = K.conv_desc(N, 128, 128, 64, 128, 128, 3, 3, 3, 1, 1, 1, 0, TG_BF16, TG_F32, ACT_NONE, 0.0, flags=1)

    def old1():
        K.conv_forward(d1, a, w1, b, None, None, t1)

    def new1():
        K.deconv_lat_forward(a, f1, b, t1)

    def old2():
        K.conv_forward(d2, t1, w2, b, None, None, t2)
        K.conv_forward(d3, t2, w3, b3, None, None, c)
        K.bicubic_add_preprocess(c, x_in, out)

    def new2():
        K.hr_tail_train(t1, f2, b, w3, b3, x_in, t2, out)

    print("forward HR tail, B = %d: conv_tran1 %6.2f -> %6.2f us | conv_tran2 + output conv + bicubic %6.2f -> %6.2f us (graph chains of 20)"
          % (N, graph_timeit(old1, chain=20), graph_timeit(new1, chain=20), graph_timeit(old2, chain=20), graph_timeit(new2, chain=20)), flush=True)
