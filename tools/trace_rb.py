"""Cycle-stamp trace of one resblock_lat workgroup (profiling tool; private -DTG_RB_TRACE build of the library).
    python tools/trace_rb.py --build   (here, cross-compiles)      python tools/trace_rb.py   (on the GPU)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tecogan_amd import build as B  # noqa: E402

so = os.path.join(ROOT, "tools", "_trace", "libtecogan_trace_rb.so")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(so), exist_ok=True)
    csrc = os.path.join(ROOT, "tecogan_amd", "csrc")
    obj = os.path.join(os.path.dirname(so), "resblock_lat_trace.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DTG_RB_TRACE", "-c", os.path.join(csrc, "resblock_lat.hip"), "-o", obj])
    others = [os.path.join(csrc, s.replace(".hip", ".o")) for s in B.SOURCES if s != "resblock_lat.hip"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
    print("built", so)
    sys.exit(0)
import tecogan_amd._lib as L  # noqa: E402

L.LIB_PATH = so
import torch  # noqa: E402

from tecogan_amd import kernels as K  # noqa: E402

lib = C.CDLL(so)
lib.tg_debug_rb_trace.argtypes = [C.POINTER(C.c_ulonglong)]
NAMES = ["entry", "first loads issued", "input region landed, LDS written", "barrier 1 passed", "level-1 MFMAs issued",
         "level-1 epilogue issued", "barrier 2 passed", "level-2 MFMAs issued", "stores issued", "stores retired (vmcnt 0)"]
N, H, W = 4, 32, 32
NB = 16
x = torch.randn(N, H, W, 64, device="cuda").bfloat16()
ws = [(torch.randn(9, 64, 64, device="cuda") * 0.03).bfloat16() for _ in range(2 * NB)]
b = torch.zeros(64, device="cuda")
aux = torch.randn(N, H, W, 64, device="cuda").bfloat16()
mid = torch.empty_like(x)
act = [torch.randn(N, H, W, 64, device="cuda").bfloat16() for _ in range(NB + 1)]
sys.path.insert(0, os.path.join(ROOT, "tools"))
from microbench import timeit  # noqa: E402

print("weight stream of the one-launch residual block [%d,%d,%d,64]: layout x prefetch distance (fragments requested ahead)" % (N, H, W))
for frag, dist in ((False, 10), (True, 4), (True, 6), (True, 8), (True, 10), (True, 14), (True, 18), (True, 36)):
    os.environ["TG_RB_DIST"] = str(dist)
    print("== %s, prefetch distance %d%s" % ("fragment-order weights" if frag else "[tap][out][in] rows", dist,
                                             " (everything up front)" if dist == 36 else ""))
    for mode, label in ((0, "forward"), (1, "input gradient")):
        def block(i, mode=mode, frag=frag):
            if mode == 0:
                K.resblock(0, act[i], ws[2 * i], b, ws[2 * i + 1], b, None, None, mid, act[i + 1], w_frag=frag)
            else:
                K.resblock(1, act[i], ws[2 * i], None, ws[2 * i + 1], None, aux, None, mid, act[i + 1], w_frag=frag)
        for i in range(NB):
            block(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(NB):
                block(i)
        us = timeit(g.replay, 30, 5) / NB
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 64)()
        assert lib.tg_debug_rb_trace(buf) == 0
        t = list(buf)
        print("  %-14s %.2f us per block in a %d-block graph chain; stamps (cycles) of the middle workgroup of the last launch:" % (label, us, NB))
        for wv in (0, 3):
            row = t[wv * 16:wv * 16 + 10]
            print("    wave %d: " % wv + "  ".join("%s +%d" % (NAMES[i][:24], row[i] - row[i - 1]) for i in range(1, 10)) +
                  "  | total %d" % (row[9] - row[0]))
