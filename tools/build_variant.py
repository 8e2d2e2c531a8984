#!/usr/bin/env python
"""Build a variant of libtecogan_hip.so with extra -D flags on ONE source (A/B of compile-time layout constants):
    python tools/build_variant.py conv3x3_ws.hip -DWS_PS=9 -> prints the .so path; run with TECOGAN_HIP_LIB=<path>."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tecogan_amd import build as B  # noqa: E402

src, defs = sys.argv[1], sys.argv[2:]
B.build(verbose=False)
out = os.path.join(ROOT, "tools", "_trace")
os.makedirs(out, exist_ok=True)
tag = src.replace(".hip", "") + "".join(d.replace("-D", "_").replace("=", "") for d in defs)
obj, so = os.path.join(out, tag + ".o"), os.path.join(out, "libtecogan_%s.so" % tag)
csrc = os.path.join(ROOT, "tecogan_amd", "csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + defs + ["-c", os.path.join(csrc, src), "-o", obj])
others = [os.path.join(csrc, s.replace(".hip", ".o")) for s in B.SOURCES if s != src]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + others)
print(so)
