"""Cycle-stamp trace of one bf16 wgrad workgroup (profiling tool; builds a private -DTG_WGRAD_TRACE copy of the library
under /tmp and points the ctypes binding at it).   python tools/trace_wgrad.py [N H W Cin Cout]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tecogan_amd import build as B
so = os.path.join(ROOT, "tools", "_trace", "libtecogan_trace.so")     # git-ignored; travels to the GPU box like the product .so
if "--build" in sys.argv:                                            # cross-compile here (no GPU needed), reusing the product objects
    os.makedirs(os.path.dirname(so), exist_ok=True)
    csrc = os.path.join(ROOT, "tecogan_amd", "csrc")
    obj = os.path.join(os.path.dirname(so), "conv_wgrad_trace.o")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-DTG_WGRAD_TRACE", "-c", os.path.join(csrc, "conv_wgrad.hip"), "-o", obj])
    others = [os.path.join(csrc, s.replace(".hip", ".o")) for s in B.SOURCES if s != "conv_wgrad.hip"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
    print("built", so)
    sys.exit(0)
import tecogan_amd._lib as L
L.LIB_PATH = so
import torch
from tools.microbench import wgrad_case
args = [a for a in sys.argv[1:] if not a.startswith("--")]
shape = tuple(int(a) for a in args[:5]) if len(args) >= 5 else (40, 32, 32, 64, 64)
fn, _ = wgrad_case(*shape)
for _ in range(5):
    fn()
torch.cuda.synchronize()
lib = C.CDLL(so)
buf = (C.c_ulonglong * 64)()
lib.tg_debug_wgrad_trace.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib.tg_debug_wgrad_trace(buf) == 0
t = list(buf)
t0 = t[0]
print("shape", shape, "PF", os.environ.get("TG_WGRAD_PF"), "blocks", os.environ.get("TG_WGRAD_BLOCKS"))
print("prologue: entry->setup %d, loads issued +%d" % (t[1] - t0, t[2] - t[1]))
i, prev = 3, t[2]
while i + 3 < 60 and t[i] > 0 and t[i] >= prev:
    print("step %2d: wait-to-top %6d | data+stage %6d | barrier %5d | loads+mfma %5d" %
          ((i - 3) // 4, t[i] - prev, t[i + 1] - t[i], t[i + 2] - t[i + 1], t[i + 3] - t[i + 2]))
    prev = t[i + 3]
    i += 4
print("loop end at +%d ; tail (atomics) %d ; total %d cycles" % (t[60] - t0, t[61] - t[60], t[61] - t0))
