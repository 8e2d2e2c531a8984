#!/usr/bin/env python
"""Workload for the HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one counter per run):
the roofline kernel of bench.py (3x3 64->64 conv @[4,32,32,64] bf16) plus a calibration copy with a known byte
count (a wide coalesced 64 MiB read + 64 MiB write), as /opt/skills/guides/MI355X_MICROARCH.md's HBM section asks.
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python tools/pmc_conv.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K                                           # noqa: E402
from tecogan_amd._lib import ACT_RELU                                          # noqa: E402

dev = torch.device("cuda", 0)
N, H, W, C = 4, 32, 32, 64
x = torch.randn(N, H, W, C, device=dev).bfloat16()
w = (torch.randn(9, C, C, device=dev) * 0.05).bfloat16()
b = torch.zeros(C, device=dev)
out = torch.empty_like(x)
d = K.conv_desc(N, H, W, C, H, W, C, 3, 3, 1, 1, 1, 0, K.dt(x), K.dt(out), ACT_RELU)
for _ in range(20):
    K.conv_forward(d, x, w, b, None, None, out)
# calibration: lincomb(a, b) -> c streams 2 x 64 MiB in and 64 MiB out (fp32, 16 B / lane)
n = 16 << 20
a0, a1, a2 = (torch.ones(n, device=dev) for _ in range(3))
for _ in range(5):
    K.lincomb(a0, a1, a2, 0.5, 0.5)
torch.cuda.synchronize()
print("pmc workload done")
