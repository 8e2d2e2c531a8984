"""Cycle-stamp trace of one conv3x3_dma workgroup (profiling tool; private -DTG_DMA_TRACE build of the library).
    python tools/trace_dma.py --build   (here, cross-compiles)      python tools/trace_dma.py   (on the GPU)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tecogan_amd import build as B  # noqa: E402

so = os.path.join(ROOT, "tools", "_trace", "libtecogan_trace_dma.so")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(so), exist_ok=True)
    csrc = os.path.join(ROOT, "tecogan_amd", "csrc")
    obj = os.path.join(os.path.dirname(so), "conv3x3_dma_trace.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DTG_DMA_TRACE", "-c", os.path.join(csrc, "conv3x3_dma.hip"), "-o", obj])
    others = [os.path.join(csrc, s.replace(".hip", ".o")) for s in B.SOURCES if s != "conv3x3_dma.hip"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
    print("built", so)
    sys.exit(0)
import tecogan_amd._lib as L  # noqa: E402

L.LIB_PATH = so
import torch  # noqa: E402

from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd._lib import ACT_RELU  # noqa: E402

lib = C.CDLL(so)
lib.tg_debug_dma_trace.argtypes = [C.POINTER(C.c_ulonglong)]
for shape in ((76, 32, 32, 256), (20, 32, 32, 256)):
    N, H, W, Cc = shape
    x = torch.randn(N, H, W, Cc, device="cuda").bfloat16()
    w = (torch.randn(9, Cc, Cc, device="cuda") * 0.05).bfloat16()
    b = torch.zeros(Cc, device="cuda")
    out = torch.empty_like(x)
    d = K.conv_desc(N, H, W, Cc, H, W, Cc, 3, 3, 1, 1, 1, 0, 1, 1, ACT_RELU)
    for _ in range(5):
        K.conv_forward(d, x, w, b, None, None, out)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    assert lib.tg_debug_dma_trace(buf) == 0
    t = list(buf)
    print("== conv3x3_dma [%d,%d,%d,%d->%d]: cycles of the middle workgroup, wave 0" % (N, H, W, Cc, Cc))
    print("  %-40s +%6d" % ("first stage's DMA issued", t[1] - t[0]))
    prev = t[1]
    lab = ["own DMA landed (vmcnt)", "barrier passed", "MFMA block + next DMA issued", "stage end (epilogue if last chunk)"]
    for it in range(15):
        for k in range(4):
            v = t[2 + 4 * it + k]
            if v < t[0]:
                break
            print("  stage %2d %-31s +%6d   (total %7d)" % (it, lab[k], v - prev, v - t[0]))
            prev = v
