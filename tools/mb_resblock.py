#!/usr/bin/env python
"""The generator's residual-block chain at the training shape ([4,32,32,64] bf16), graph-chained dependent launches:
one launch per block (csrc/resblock_lat.hip) against the two conv3x3_tile launches it replaces, forward and input-gradient
form.  Prints microseconds per BLOCK."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd._lib import ACT_NONE, ACT_RELU, TG_BF16  # noqa: E402
from tools.microbench import timeit  # noqa: E402

DEV = "cuda"
torch.manual_seed(0)
NB = 16


def chain_time(step_fn, iters=30):
    """step_fn(i) enqueues block i; the graph holds NB dependent blocks (different weights per block, as in the generator)."""
    for i in range(NB):
        step_fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(NB):
            step_fn(i)
    return timeit(g.replay, iters, 5) / NB


for N, H, W in ((4, 32, 32), (4, 24, 24), (1, 32, 32), (8, 32, 32)) + (((1, 270, 480),) if "--big" in sys.argv else ()):
    bf = torch.bfloat16
    a = [torch.randn(N, H, W, 64, device=DEV).to(bf) * 0.5 for _ in range(NB + 1)]
    r = [torch.empty(N, H, W, 64, device=DEV, dtype=bf) for _ in range(NB)]
    w1 = [(torch.randn(9, 64, 64, device=DEV) * 0.03).to(bf) for _ in range(NB)]
    w2 = [(torch.randn(9, 64, 64, device=DEV) * 0.03).to(bf) for _ in range(NB)]
    b1 = [torch.zeros(64, device=DEV) for _ in range(NB)]
    d1 = K.conv_desc(N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16, ACT_RELU)
    d2 = K.conv_desc(N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 0, TG_BF16, TG_BF16, ACT_NONE)
    dA = K.conv_desc(N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 1, TG_BF16, TG_BF16, 0, 0.0, ACT_RELU, 0.0)
    dB = K.conv_desc(N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 1, TG_BF16, TG_BF16, 0, 0.0, ACT_NONE, 0.0)

    def two_fwd(i):
        K.conv_forward(d1, a[i], w1[i], b1[i], None, None, r[i])
        K.conv_forward(d2, r[i], w2[i], b1[i], a[i], None, a[i + 1])

    f1, f2 = [K.frag_order(w) for w in w1], [K.frag_order(w) for w in w2]

    def one_fwd(i):
        K.resblock(0, a[i], f1[i], b1[i], f2[i], b1[i], None, None, r[i], a[i + 1], w_frag=True)

    aux = [torch.randn(N, H, W, 64, device=DEV).to(bf) for _ in range(NB)]
    gmid = [torch.empty(N, H, W, 64, device=DEV, dtype=bf) for _ in range(NB)]

    def two_bwd(i):
        K.conv_forward(dA, a[i], w2[i], None, None, aux[i], gmid[i])
        K.conv_forward(dB, gmid[i], w1[i], None, a[i], None, a[i + 1])

    def one_bwd(i):
        K.resblock(1, a[i], f2[i], None, f1[i], None, aux[i], None, gmid[i], a[i + 1], w_frag=True)

    t2f, t1f, t2b, t1b = chain_time(two_fwd), chain_time(one_fwd), chain_time(two_bwd), chain_time(one_bwd)
    print("res block [%d,%d,%d,64] bf16, us per block in a %d-block graph chain: forward two launches %6.2f  one launch %6.2f | "
          "input gradient two launches %6.2f  one launch %6.2f" % (N, H, W, NB, t2f, t1f, t2b, t1b), flush=True)
