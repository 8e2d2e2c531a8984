#!/usr/bin/env python
"""conv3x3_ws at its throughput shapes: the 1080p inference res-block convs [1,270,480,64] (plain / +residual), VGG conv1_2
[76,128,128,64] and its input gradient form.  Run twice with TECOGAN_HIP_LIB pointing at a layout variant for an A/B."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd._lib import ACT_NONE, ACT_RELU  # noqa: E402
from tools.microbench import graph_timeit  # noqa: E402

tag = os.path.basename(os.environ.get("TECOGAN_HIP_LIB", "default"))
for N, H, W in ((1, 270, 480), (76, 128, 128), (20, 128, 128)):
    x = torch.randn(N, H, W, 64, device="cuda").bfloat16()
    r = torch.randn(N, H, W, 64, device="cuda").bfloat16()
    w = (torch.randn(9, 64, 64, device="cuda") * 0.05).bfloat16()
    b = torch.zeros(64, device="cuda")
    out = torch.empty_like(x)
    d0 = K.conv_desc(N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 0, 1, 1, ACT_RELU)
    d1 = K.conv_desc(N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 0, 1, 1, ACT_NONE)
    ta = graph_timeit(lambda: K.conv_forward(d0, x, w, b, None, None, out))
    tb = graph_timeit(lambda: K.conv_forward(d1, x, w, b, r, None, out))
    wfr = K.frag_order(w)
    tc = td = float("nan")
    if hasattr(K, "conv3x3_c64_frag"):
        tc = graph_timeit(lambda: K.conv3x3_c64_frag(x, wfr, b, None, out, ACT_RELU))
        td = graph_timeit(lambda: K.conv3x3_c64_frag(x, wfr, b, r, out, ACT_NONE))
    te = float("nan")
    if hasattr(K, "resblock_c64_thr"):          # the whole residual block as one launch (csrc/resblock_thr.hip)
        out2 = torch.empty_like(x)
        te = graph_timeit(lambda: K.resblock_c64_thr(x, wfr, b, wfr, b, out2))
    fl = 2.0 * N * H * W * 64 * 64 * 9
    print("[%s] residual block [%d,%d,%d]: two launches (fragment-order weights) %6.1f us | one launch %6.1f us (%4.0f TFLOP/s of 2 convs)"
          % (tag, N, H, W, tc + td, te, 2 * fl / te * 1e-6), flush=True)
    print("[%s] conv 64->64 [%d,%d,%d]: relu %6.1f us (%4.0f TFLOP/s)   +residual %6.1f us | fragment-order weights (per CU %s): %6.1f us  +residual %6.1f us"
          % (tag, N, H, W, ta, fl / ta * 1e-6, tb, "2", tc, td), flush=True)
