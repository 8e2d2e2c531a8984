"""Per-node floor of a dependent kernel chain inside a hipGraph: a 64-element lincomb (one workgroup)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K
from tools.microbench import graph_timeit
a = torch.ones(64, device="cuda"); b = torch.ones(64, device="cuda")
print("graph node floor (tiny dependent kernel): %.2f us" % graph_timeit(lambda: K.lincomb(a, b, a, 0.5, 0.5)))
n = 4 * 32 * 32 * 64
x = torch.ones(n, device="cuda"); y = torch.ones(n, device="cuda")
print("1 MiB fp32 lincomb node: %.2f us" % graph_timeit(lambda: K.lincomb(x, y, x, 0.5, 0.5)))
