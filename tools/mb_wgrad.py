import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import wgrad_case, graph_timeit
for name, args in (("gen40", (40, 32, 32, 64, 64)), ("tran2", (40, 128, 128, 64, 64, 3, 2)), ("fnet36x32", (36, 32, 32, 32, 32)), ("fnet256", (36, 4, 4, 256, 256)), ("vgg-like", (24, 128, 128, 64, 64, 4, 2))):
    fn, flops = wgrad_case(*args)
    t = graph_timeit(fn, chain=20)
    print("wgrad %-10s blocks=%s: %.2f us  %.1f TFLOP/s" % (name, os.environ.get("TG_WGRAD_BLOCKS"), t, flops / t / 1e6))
