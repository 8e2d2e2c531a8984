"""Do transitions between different kernels inside a hipGraph cost extra?  Alternating chains vs the sum of homogeneous ones."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import conv_case, wgrad_case, graph_timeit
a, _ = conv_case(4, 32, 32, 64, 64)
b, _ = conv_case(4, 128, 128, 8, 64)
c, _ = wgrad_case(40, 32, 32, 64, 64)
ta, tb, tc = (graph_timeit(f, chain=100, iters=10) for f in (a, b, c))
print("alone: LR conv %.2f  c8 conv %.2f  wgrad %.2f us" % (ta, tb, tc))
for name, f, g, s in (("LRconv+c8", a, b, ta + tb), ("LRconv+wgrad", a, c, ta + tc)):
    def pair():
        f(); g()
    print("%-14s alternating %.2f us per pair vs %.2f summed" % (name, graph_timeit(pair, chain=100, iters=10), s))
