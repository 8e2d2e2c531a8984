#!/usr/bin/env python
"""FNet forward + backward alone (graph-replayed), per launch type: where do the 0.8 ms of the backward pass go?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import OrderedDict
from tecogan_amd import kernels as K
from tecogan_amd.nets import FNET_CPAD, FNet
from tecogan_amd.params import ParamStore, fnet_spec, init_values
from tools.microbench import timeit

N = int(os.environ.get("MB_N", "72"))
ps = ParamStore(OrderedDict(fnet=fnet_spec()), torch.device("cuda"), torch.bfloat16,
                bpad=("fnet/autoencode_unit/output_stage/conv2/Conv/weights",))       # as TrainEngine builds it
ps.load(init_values(fnet_spec(), 1))
fn = FNet(ps)
x = torch.rand(N, 32, 32, FNET_CPAD, device="cuda").bfloat16()
flow, saved = fn.forward(x)
dflow = torch.randn_like(flow)
for _ in range(3):
    fn.backward(saved, dflow)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fn.backward(saved, dflow)
t = timeit(g.replay, 50, 5)
K.prof_collect(); K.prof_enable(True); fn.backward(saved, dflow); torch.cuda.synchronize(); K.prof_enable(False)
ents = K.prof_collect()
print("FNet backward N=%d multi=%s target=%s: %.1f us per pass; instrumented launches: %s" % (
    N, "1", "-", t,
    ", ".join("%s x%d %.0f us" % (e["name"], e["calls"], e["total_us"]) for e in ents[:6])))
for e in ents:
    if e["name"].startswith("conv_wgrad"):
        print("   %-32s x%d %.0f us  %.1f MFLOP  %.2f MB" % (e["name"], e["calls"], e["total_us"], e["flops"] / 1e6, e["bytes"] / 1e6))
tot = sum(e["total_us"] for e in ents)
print("   all instrumented launches of one backward pass (%.0f us of kernel time in %d launches):" % (tot, sum(e["calls"] for e in ents)))
for e in ents:
    print("   %-40s x%-3d %7.1f us  (%5.1f us each)  %8.1f MFLOP" % (e["name"], e["calls"], e["total_us"], e["total_us"] / e["calls"], e["flops"] / 1e6))
# forward pass, the same way
for _ in range(3):
    fn.forward(x)
torch.cuda.synchronize()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    fn.forward(x)
t2 = timeit(g2.replay, 50, 5)
K.prof_collect(); K.prof_enable(True); fn.forward(x); torch.cuda.synchronize(); K.prof_enable(False)
ents = K.prof_collect()
print("FNet forward N=%d: %.1f us per pass" % (N, t2))
for e in ents:
    print("   %-40s x%-3d %7.1f us  (%5.1f us each)  %8.1f MFLOP" % (e["name"], e["calls"], e["total_us"], e["total_us"] / e["calls"], e["flops"] / 1e6))
