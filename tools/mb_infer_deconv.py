#!/usr/bin/env python
"""conv_tran1 of the inference step ([1,270,480,64] -> [1,540,960,64], ReLU): the throughput-regime phase kernel
(deconv3x3s2_ws, csrc/conv3x3_ws.hip) against the latency-regime one of the training recurrence (hr_fwd_lat<deconv>)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd._lib import ACT_RELU  # noqa: E402
from tools.microbench import graph_timeit  # noqa: E402

bf = torch.bfloat16
for h, w in ((270, 480), (144, 180)):
    x = torch.randn(1, h, w, 64, device="cuda").to(bf)
    wt = (torch.randn(9, 64, 64, device="cuda") * 0.05).to(bf)          # [tap][out][in] = TF's [kh,kw,Cout,Cin]
    b = torch.zeros(64, device="cuda")
    y = torch.empty(1, 2 * h, 2 * w, 64, device="cuda", dtype=bf)
    y2 = torch.empty_like(y)
    d = K.conv_desc(1, h, w, 64, 2 * h, 2 * w, 64, 3, 3, 2, 0, 0, 1, 1, 1, ACT_RELU)
    ta = graph_timeit(lambda: K.conv_forward(d, x, wt, b, None, None, y), chain=20)
    wf = K.frag_order(wt)
    tb = graph_timeit(lambda: K.deconv_lat_forward(x, wf, b, y2), chain=20)
    err = (y.float() - y2.float()).abs().max().item()
    print("conv_tran1 [1,%d,%d,64]: throughput kernel %6.1f us   latency kernel %6.1f us   (max |diff| %.3g)" % (h, w, ta, tb, err), flush=True)
