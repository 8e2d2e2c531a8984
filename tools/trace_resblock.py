"""Cycle-stamp trace of one fused residual-block workgroup (profiling tool; private -DTG_RES_TRACE build of the library).
    python tools/trace_resblock.py --build   (here, cross-compiles)      python tools/trace_resblock.py   (on the GPU)"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tecogan_amd import build as B
so = os.path.join(ROOT, "tools", "_trace", "libtecogan_trace_res.so")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(so), exist_ok=True)
    csrc = os.path.join(ROOT, "tecogan_amd", "csrc")
    obj = os.path.join(os.path.dirname(so), "resblock_trace.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DTG_RES_TRACE", "-c", os.path.join(csrc, "resblock.hip"), "-o", obj])
    others = [os.path.join(csrc, s.replace(".hip", ".o")) for s in B.SOURCES if s != "resblock.hip"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
    print("built", so)
    sys.exit(0)
import tecogan_amd._lib as L
L.LIB_PATH = so
import torch
from tecogan_amd import kernels as K
N, H, W = 4, 32, 32
x = torch.randn(N, H, W, 64, device="cuda").bfloat16()
w1 = (torch.randn(9, 64, 64, device="cuda") * 0.05).bfloat16(); w2 = w1.clone()
b = torch.zeros(64, device="cuda")
mid, out = torch.empty_like(x), torch.empty_like(x)
for _ in range(5):
    K.resblock_fused(x, w1, b, None, mid, w2, b, None, out, False, True)
torch.cuda.synchronize()
lib = C.CDLL(so)
buf = (C.c_ulonglong * 16)()
lib.tg_debug_res_trace.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib.tg_debug_res_trace(buf) == 0
t = list(buf)
names = ["entry", "loads issued", "x staged + barrier", "stage-1 MFMAs issued", "stage-1 epilogue + barrier", "mid stores issued",
         "stage-2 MFMAs issued", "stage-2 epilogue + barrier", "out stores issued", "stores retired"]
for i in range(1, 10):
    print("%-28s +%6d   (total %6d)" % (names[i], t[i] - t[i - 1], t[i] - t[0]))
