"""Does the per-node cost of a dependent kernel chain grow with the length of the hipGraph?  (bubble hunt)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import conv_case, graph_timeit
fn, _ = conv_case(4, 32, 32, 64, 64)
for chain in (50, 200, 600, 1500, 3000):
    print("chain %5d nodes: %.3f us per conv node" % (chain, graph_timeit(fn, chain=chain, iters=10)))
