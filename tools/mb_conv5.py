#!/usr/bin/env python
"""VGG conv5_x (8x8 images, 512 -> 512; packed 16x16 tiles of 4 images) at the image counts of the training step's VGG passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd._lib import ACT_RELU  # noqa: E402
from tools.microbench import graph_timeit  # noqa: E402

tag = os.path.basename(os.environ.get("TECOGAN_HIP_LIB", "default"))
for N in (28, 32, 44, 48, 76):
    x = torch.randn(N, 8, 8, 512, device="cuda").bfloat16()
    w = (torch.randn(9, 512, 512, device="cuda") * 0.02).bfloat16()
    b = torch.zeros(512, device="cuda")
    out = torch.empty_like(x)
    d = K.conv_desc(N, 8, 8, 512, 8, 8, 512, 3, 3, 1, 1, 1, 0, 1, 1, ACT_RELU)
    t = graph_timeit(lambda: K.conv_forward(d, x, w, b, None, None, out), chain=20 if "--pmc" in sys.argv else 50, iters=2 if "--pmc" in sys.argv else 20)
    print("[%s] conv5 [%d,8,8,512->512]: %6.1f us (%4.0f TFLOP/s)" % (tag, N, t, 2.0 * N * 64 * 512 * 512 * 9 / t * 1e-6), flush=True)
