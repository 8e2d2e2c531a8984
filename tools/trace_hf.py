"""Cycle stamps of the fused HR tail (csrc/hr_fwd_lat.hip) at the 1080p inference shape, both tile forms: where a workgroup's time goes.
    python tools/build_variant.py hr_fwd_lat.hip -DTG_HF_TRACE      (here)
    TECOGAN_HIP_LIB=tools/_trace/libtecogan_hr_fwd_lat_TG_HF_TRACE.so python tools/trace_hf.py      (GPU)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tecogan_amd import _lib as L  # noqa: E402
from tecogan_amd import kernels as K  # noqa: E402
from tools.microbench import graph_timeit  # noqa: E402

DEV = "cuda"
N, h2, w2 = (4, 64, 64) if "--train" in sys.argv else (1, 540, 960)      # t1 of the training recurrence / of the 1080p frame
t1 = torch.randn(N, h2, w2, 64, device=DEV).bfloat16()
f2 = K.frag_order((torch.randn(9, 64, 64, device=DEV) * 0.06).bfloat16())
w3 = (torch.randn(9, 3, 64, device=DEV) * 0.06).bfloat16()
bt, bo = torch.randn(64, device=DEV) * 0.1, torch.randn(3, device=DEV) * 0.1
gen_in = torch.randn(N, h2 // 2, w2 // 2, 56, device=DEV).bfloat16()
st = torch.empty(N, 2 * h2, 2 * w2, 3, device=DEV)
lib = C.CDLL(L.LIB_PATH)
names = ["region load + barrier", "phase 0 MFMAs", "epilogue", "phase 1 MFMAs", "epilogue", "phase 2 MFMAs", "epilogue", "phase 3 MFMAs",
         "epilogue", "w3 + barrier", "output conv"]
for tile in (("4",) if "--train" in sys.argv else ("4", "8")):
    os.environ["TG_HR_TAIL_TILE"] = tile
    us = graph_timeit(lambda: K.hr_tail_train(t1, f2, bt, w3, bo, gen_in, None, None, st), 10, 5)
    print("tile form %s x %s: %.1f us per launch" % ((("4", "8") if tile == "4" else ("8", "16")) + (us,)))
    if hasattr(lib, "tg_debug_hf_trace"):
        buf = (C.c_ulonglong * 64)()
        lib.tg_debug_hf_trace.argtypes = [C.POINTER(C.c_ulonglong)]
        assert lib.tg_debug_hf_trace(buf) == 0
        t = list(buf)
        for wv in (0, 3):
            r = t[wv * 16:wv * 16 + 12]
            print("   wave %d: " % wv + "  ".join("%s %d" % (names[i], r[i + 1] - r[i]) for i in range(11)) + "  | workgroup %d cycles" % (r[11] - r[0]))
