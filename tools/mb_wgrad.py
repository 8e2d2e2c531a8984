import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import wgrad_case, graph_timeit
for name, args in (("gen40", (40, 32, 32, 64, 64)), ("tran2", (40, 128, 128, 64, 64, 3, 2)), ("fnet36x32", (36, 32, 32, 32, 32)), ("fnet256", (36, 4, 4, 256, 256)), ("vgg-like", (24, 128, 128, 64, 64, 4, 2))):
    fn, flops = wgrad_case(*args)
    t = graph_timeit(fn, chain=20)
    print("wgrad %-10s blocks=%s: %.2f us  %.1f TFLOP/s" % (name, "auto", t, flops / t / 1e6))

# the generator trunk's grouped launch as the TecoGAN / FRVSR steps issue it: 2*nres layers, T*B images of 32x32x64
from tecogan_amd import kernels as K
for G, N in ((32, 76), (20, 40)):
    d = K.conv_desc(N, 32, 32, 64, 32, 32, 64, 3, 3, 1, 1, 1, 0, 0, 0)
    xs = [torch.randn(N, 32, 32, 64, device="cuda").bfloat16() for _ in range(G)]
    ys = [torch.randn(N, 32, 32, 64, device="cuda").bfloat16() for _ in range(G)]
    dws = [torch.zeros(3, 3, 64, 64, device="cuda") for _ in range(G)]
    dbs = [torch.zeros(64, device="cuda") for _ in range(G)]
    fn = lambda: K.conv_wgrad_grouped(d, xs, ys, dws, dbs, ldx=64, ldy=64)
    t = graph_timeit(fn, chain=5)
    fl = 2.0 * G * N * 32 * 32 * 64 * 64 * 9
    print("wgrad grouped G=%d N=%d (TG_WGRAD_TR=%s): %.1f us  %.1f TFLOP/s" % (G, N, os.environ.get("TG_WGRAD_TR"), t, fl / t / 1e6))

# the generator's output conv (64 -> 3, gradient tensor padded to 8 channels) over T*B = 76 / 40 HR frames
for N in (76, 40):
    d = K.conv_desc(N, 128, 128, 64, 128, 128, 3, 3, 3, 1, 1, 1, 0, 0, 0)
    x = torch.randn(N, 128, 128, 64, device="cuda").bfloat16()
    y = torch.randn(N, 128, 128, 8, device="cuda").bfloat16()
    dw = torch.zeros(3, 3, 64, 3, device="cuda")
    db = torch.zeros(3, device="cuda")
    fn = lambda: K.conv_wgrad(d, x, y, dw, db, ldx=64, ldy=8)
    t = graph_timeit(fn, chain=5)
    print("wgrad out-conv N=%d (TG_WGRAD_TR=%s): %.1f us  %.2f TB/s" % (N, os.environ.get("TG_WGRAD_TR"), t, N * 16384 * 144 / t / 1e6))
