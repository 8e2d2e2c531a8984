#!/usr/bin/env python
"""The inference step's HR tail at 1080p (t1 = [1,540,960,64]): the throughput-regime kernel (csrc/hr_tail.hip) against the
latency-regime training kernel without the t2 store (csrc/hr_fwd_lat.hip: small tiles, weights re-streamed per tile, 4
workgroups per CU)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tools.microbench import timeit  # noqa: E402

DEV = "cuda"
torch.manual_seed(0)
bf = torch.bfloat16
for h2, w2 in ((540, 960), (288, 360)):
    t1 = torch.randn(1, h2, w2, 64, device=DEV).to(bf)
    w2t = (torch.randn(9, 64, 64, device=DEV) * 0.05).to(bf)
    w3 = (torch.randn(9, 3, 64, device=DEV) * 0.05).to(bf)
    b, b3 = torch.zeros(64, device=DEV), torch.zeros(3, device=DEV)
    x_in = torch.randn(1, h2 // 2, w2 // 2, 56, device=DEV).to(bf)
    out = torch.empty(1, 2 * h2, 2 * w2, 3, device=DEV)
    st = torch.empty_like(out)
    f2 = K.frag_order(w2t)
    ta = timeit(lambda: K.hr_tail_forward(t1, w2t, b, w3, b3, x_in, out, None), 50, 5)
    tb = timeit(lambda: K.hr_tail_train(t1, f2, b, w3, b3, x_in, None, out), 50, 5)
    tc = timeit(lambda: K.hr_tail_train(t1, f2, b, w3, b3, x_in, None, None, st), 50, 5)
    print("HR tail, t1 [1,%d,%d,64]: hr_tail %7.1f us   hr_fwd_lat<tail> without t2 store %7.1f us (state only %7.1f us)"
          % (h2, w2, ta, tb, tc), flush=True)
