import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tecogan_amd import kernels as K
from tools.microbench import graph_timeit, conv_case, timeit
N, H, W = 4, 32, 32
x = torch.randn(N, H, W, 64, device="cuda").bfloat16()
w1 = (torch.randn(9, 64, 64, device="cuda") * 0.05).bfloat16(); w2 = w1.clone()
b = torch.zeros(64, device="cuda")
mid, out, m = torch.empty_like(x), torch.empty_like(x), torch.randn_like(x.float()).bfloat16()
fwd = lambda: K.resblock_fused(x, w1, b, None, mid, w2, b, None, out, False, True)
bwd = lambda: K.resblock_fused(x, w1, None, m, mid, w2, None, m, out, True, False)
print("fused fwd: %.2f us (graph chain)  %.2f us (eager)" % (graph_timeit(fwd), timeit(fwd)))
print("fused bwd: %.2f us (graph chain)" % graph_timeit(bwd))
fn, _ = conv_case(4, 32, 32, 64, 64)
print("single conv3x3: %.2f us (graph chain)" % graph_timeit(fn))
# alternating chain like the real generator: conv -> conv with data dependence
a = x.clone(); c = torch.empty_like(x)
d = K.conv_desc(N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 0, K.dt(x), K.dt(x), 1)
def pair():
    K.conv_forward(d, a, w1, b, None, None, c)
    K.conv_forward(d, c, w2, b, a, None, mid)
print("dependent conv pair: %.2f us (graph chain)" % graph_timeit(pair))
def fpair():
    K.resblock_fused(a, w1, b, None, c, w2, b, None, mid, False, True)
    K.resblock_fused(mid, w1, b, None, c, w2, b, None, a, False, True)
print("2 dependent fused blocks: %.2f us (graph chain)" % graph_timeit(fpair))
