#!/usr/bin/env python
"""Is the spread of the discriminator's weight-gradient error at configs[2] (fp32 mode, against the fp64 oracle) a property
of fp32 summation order or of the two-stream schedule?  One oracle step, then several engine steps from the same weights and
batch under different schedules (environment read at engine construction), worst tensors printed per run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from oracle import teco as OT  # noqa: E402
from tecogan_amd.engine import TrainEngine  # noqa: E402
from tecogan_amd.params import damp_values  # noqa: E402
from test_train_gpu import make_batch  # noqa: E402

F = OT.default_flags()
S = OT.State(F, seed=42, gan=True, dtype=torch.float64)
S.P = damp_values(S.P)
P0 = {k: v.clone() for k, v in S.P.items()}
x, y = make_batch(F.batch_size, F.RNN_N, F.crop_size)
R = OT.train_step(S, x.double(), y.double())
print("oracle step done", flush=True)
for tag, env in (("default", {}), ("default", {}), ("serial", {"TG_OVERLAP": "0"}), ("serial", {"TG_OVERLAP": "0"}),
                 ("no-down-on-side", {"TG_OVERLAP_PARTS": "39"}), ("default", {})):
    for k in ("TG_OVERLAP", "TG_OVERLAP_PARTS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    eng = TrainEngine(F, "cuda:0", gan=True, act_dtype=torch.float32, seed=7, use_graph=False)
    eng.ps.load(P0)
    eng.vps.load(S.vgg)
    eng.step(x.cuda(), y.cuda())
    torch.cuda.synchronize()
    rows = []
    for name, g in R["grads"].items():
        mine = eng.ps.gview(name).detach().cpu().double()
        ref = g.detach().double()
        l2 = ((mine - ref).norm() / ref.norm().clamp_min(1e-30)).item()
        mx = ((mine - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
        rows.append((mx, l2, name))
    rows.sort(reverse=True)
    print("%-16s worst max-norm: %s" % (tag, "; ".join("%s %.2e (L2 %.2e)" % (n.split("/")[-4] + "/" + n.split("/")[-3], m, l) for m, l, n in rows[:3])),
          flush=True)
    del eng
    torch.cuda.empty_cache()
