"""Cycle-stamp trace of one conv3x3_ws workgroup (profiling tool; private -DTG_WS_TRACE build of the library).
    python tools/trace_ws.py --build   (here, cross-compiles)      python tools/trace_ws.py   (on the GPU)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tecogan_amd import build as B  # noqa: E402

so = os.path.join(ROOT, "tools", "_trace", "libtecogan_trace_ws.so")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(so), exist_ok=True)
    csrc = os.path.join(ROOT, "tecogan_amd", "csrc")
    obj = os.path.join(os.path.dirname(so), "conv3x3_ws_trace.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DTG_WS_TRACE", "-c", os.path.join(csrc, "conv3x3_ws.hip"), "-o", obj])
    others = [os.path.join(csrc, s.replace(".hip", ".o")) for s in B.SOURCES if s != "conv3x3_ws.hip"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
    print("built", so)
    sys.exit(0)
import tecogan_amd._lib as L  # noqa: E402

L.LIB_PATH = so
import torch  # noqa: E402

from tecogan_amd import kernels as K  # noqa: E402
from tecogan_amd._lib import ACT_RELU  # noqa: E402

lib = C.CDLL(so)
lib.tg_debug_ws_trace.argtypes = [C.POINTER(C.c_ulonglong)]
for shape in ((1, 270, 480), (76, 128, 128)):
    N, H, W = shape
    x = torch.randn(N, H, W, 64, device="cuda").bfloat16()
    w = (torch.randn(9, 64, 64, device="cuda") * 0.05).bfloat16()
    b = torch.zeros(64, device="cuda")
    out = torch.empty_like(x)
    d = K.conv_desc(N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 0, 1, 1, ACT_RELU)
    for _ in range(5):
        K.conv_forward(d, x, w, b, None, None, out)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    assert lib.tg_debug_ws_trace(buf) == 0
    t = list(buf)
    print("== conv3x3_ws [%d,%d,%d,64->64]: cycles of the middle workgroup, wave 0" % shape)
    names = ["entry", "first DMA issued", "weight loads issued", "first wait (DMA + weights landed)"]
    for i in range(1, 4):
        print("  %-36s +%6d   (total %6d)" % (names[i], t[i] - t[i - 1], t[i] - t[0]))
    it = 0
    while 4 + 5 * it < 63 and t[4 + 5 * it] > t[3] and it < 8:
        base = 4 + 5 * it
        prev = t[base - 1] if it else t[3]
        lab = ["barrier passed", "next DMA issued", "MFMA block issued", "epilogue stores issued", "vmcnt(8): next tile landed"]
        for k in range(5):
            if t[base + k] == 0 or t[base + k] < t[0]:
                break
            print("  tile %d %-28s +%6d   (total %6d)" % (it, lab[k], t[base + k] - prev, t[base + k] - t[0]))
            prev = t[base + k]
        it += 1
    print("  %-36s            (total %6d)" % ("all stores retired", t[63] - t[0]))
