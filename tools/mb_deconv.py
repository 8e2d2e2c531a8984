#!/usr/bin/env python
"""The generator's two transposed convs (k3 s2, 64 -> 64, ReLU) at the TRAINING chain's shapes, graph-chained launches:
which kernel serves them (conv_igemm below 256 input tiles, deconv3x3s2_ws from there on) and what it costs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd._lib import ACT_RELU  # noqa: E402
from tools.microbench import graph_timeit  # noqa: E402

DEV = "cuda"
torch.manual_seed(0)
for N, H in ((4, 32), (4, 64), (1, 270)):
    W = H if H != 270 else 480
    x = torch.randn(N, H, W, 64, device=DEV).bfloat16()
    w = (torch.randn(9, 64, 64, device=DEV) * 0.05).bfloat16()          # [tap][in][out] natural copy
    b = torch.zeros(64, device=DEV)
    out = torch.empty(N, 2 * H, 2 * W, 64, device=DEV, dtype=torch.bfloat16)
    d = K.conv_desc(N, H, W, 64, 2 * H, 2 * W, 64, 3, 3, 2, 0, 0, 1, K.dt(x), K.dt(out), ACT_RELU, 0.0)
    K.prof_collect(); K.prof_enable(True); K.conv_forward(d, x, w, b, None, None, out); torch.cuda.synchronize(); K.prof_enable(False)
    name = K.prof_collect()[0]["name"]
    t = graph_timeit(lambda: K.conv_forward(d, x, w, b, None, None, out), chain=20)
    print("deconv k3 s2 [%d,%d,%d,64->64]  %-24s %7.2f us" % (N, H, W, name, t), flush=True)
