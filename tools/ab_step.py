"""Same-box A/B of engine settings on the timed training steps (configs[2] TecoGAN, configs[1] FRVSR; hipGraph replay):
    python tools/ab_step.py "name=<python statements on `eng`>" ...  [--rounds 2] [--steps 100] [--configs tecogan,frvsr]
Every variant builds a fresh engine, applies its statements (`eng`, `K`, `params`, `nets` in scope; empty = the defaults) before
the first step, and is timed `rounds` times in alternation with the others.  Example:
    python tools/ab_step.py "per-block=eng.G.resblock_chain=False" "chain="
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench as B  # noqa: E402
from tecogan_amd import kernels as K, nets, params  # noqa: E402,F401

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opt = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
rounds, steps = int(opt.get("--rounds", 2)), int(opt.get("--steps", 100))
configs = opt.get("--configs", "tecogan,frvsr").split(",")
variants = [(a.split("=", 1)[0], a.split("=", 1)[1]) for a in args]
dev = torch.device("cuda", 0)
res = {}
for config in configs:
    for r in range(rounds):
        for name, code in variants:
            eng = B.new_engine(config, "bf16", dev)
            exec(code, {"eng": eng, "K": K, "params": params, "nets": nets, "torch": torch})
            x, y = B.synthetic_batch(eng.F, 1234, dev)
            eng.set_batch(x, y)
            dt = B.time_steps(eng, steps, 10, torch.cuda.synchronize)
            res.setdefault((config, name), []).append(dt / steps * 1e3)
            del eng
            torch.cuda.empty_cache()
            time.sleep(0.2)
for config in configs:
    print("%s step (ms), %d steps per run, runs in alternation:" % (config, steps))
    for name, code in variants:
        v = res[(config, name)]
        print("  %-28s %s   [%s]" % (name, "  ".join("%.3f" % t for t in v), code or "defaults"))
