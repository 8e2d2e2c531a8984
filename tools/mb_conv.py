import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import conv_case, graph_timeit, timeit
shapes = {"gen": (4, 32, 32, 64, 64), "inf": (1, 270, 480, 64, 64), "vgg1": (76, 128, 128, 64, 64), "vgg3": (76, 32, 32, 256, 256), "fnet": (36, 16, 16, 64, 64),
          "c8": (4, 128, 128, 8, 64), "c8vgg": (76, 128, 128, 8, 64), "out": (4, 128, 128, 64, 8)}
name = sys.argv[1]
fn, flops = conv_case(*shapes[name])
t = graph_timeit(fn)
print("%s force=%s: %.2f us  %.1f TFLOP/s" % (name, os.environ.get("TG_C3_FORCE"), t, flops / t / 1e6))
