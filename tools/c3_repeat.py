#!/usr/bin/env python
"""Run-to-run spread of the fp32 parity mode: several engine steps from the same weights and batch, each a fresh engine, with
a sha256 of everything the step produces (HR frames, loss slots, the flat gradient buffer, the post-Adam weights).

    python tools/c3_repeat.py [--config c3|small] [--runs 6] [--oracle]

Default mode (fp32 atomics: split-K weight gradients, scatter kernels, loss / batch-norm reductions): the digests differ from
run to run and the discriminator's gradient error against the fp64 oracle is bimodal (profiles/r02z_c3_repeat.txt).
TG_DETERMINISTIC=1 (csrc/common.h: ordered reductions -- one workgroup per reduction, no split-K, scatter kernels as one
wavefront): every run must print the SAME digest; the last line says IDENTICAL or DIFFERENT.  --oracle adds the per-run worst
gradient errors against ONE fp64 oracle step (config c3: ~3 minutes of CPU)."""
import argparse
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from oracle import teco as OT  # noqa: E402
from tecogan_amd.engine import TrainEngine  # noqa: E402
from tecogan_amd.params import damp_values  # noqa: E402


def ratios(blob, out):
    """One deterministic-mode C3 step from the weights / batch / oracle gradients the parity test saved: per gradient tensor
    [relative L2, max-norm] of this path against the fp64 oracle, and the fp32 oracle's own two figures."""
    import json
    d = torch.load(blob)
    F = OT.default_flags()
    eng = TrainEngine(F, "cuda:0", gan=True, act_dtype=torch.float32, seed=7, use_graph=False)
    eng.ps.load(d["P_init"])
    eng.vps.load(d["vgg"])
    eng.step(d["x"].cuda(), d["y"].cuda())
    torch.cuda.synchronize()
    res = {}
    for name, ref in d["g64"].items():
        ref = ref.double()
        mine, o32 = eng.ps.gview(name).detach().cpu().double(), d["g32"][name].double()
        nrm, mxn = ref.norm().clamp_min(1e-30), ref.abs().max().clamp_min(1e-30)
        res[name] = [((mine - ref).norm() / nrm).item(), ((mine - ref).abs().max() / mxn).item(),
                     ((o32 - ref).norm() / nrm).item(), ((o32 - ref).abs().max() / mxn).item()]
    json.dump(res, open(out, "w"), indent=0, sort_keys=True)


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--ratios":
        assert os.environ.get("TG_DETERMINISTIC") == "1", "the frozen numbers belong to the ordered-reduction parity mode"
        return ratios(sys.argv[2], sys.argv[3])
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=["c3", "small"])
    ap.add_argument("--runs", type=int, default=6)
    ap.add_argument("--oracle", action="store_true")
    a = ap.parse_args()
    F = OT.default_flags() if a.config == "c3" else OT.default_flags(batch_size=2, RNN_N=3, crop_size=16, num_resblock=2)
    S = OT.State(F, seed=42, gan=True, dtype=torch.float64)
    S.P = damp_values(S.P)
    P0 = {k: v.clone() for k, v in S.P.items()}
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(F.batch_size, F.RNN_N, F.crop_size, F.crop_size, 3, generator=g)
    y = torch.rand(F.batch_size, F.RNN_N, 4 * F.crop_size, 4 * F.crop_size, 3, generator=g) * 2 - 1
    R = OT.train_step(S, x.double(), y.double()) if a.oracle else None
    print("mode: %s  TG_DETERMINISTIC=%s" % (a.config, os.environ.get("TG_DETERMINISTIC", "0")), flush=True)
    digests = []
    for run in range(a.runs):
        eng = TrainEngine(F, "cuda:0", gan=True, act_dtype=torch.float32, seed=7, use_graph=False)
        eng.ps.load(P0)
        eng.vps.load(S.vgg)
        eng.step(x.cuda(), y.cuda())
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for t in (eng.gen, eng.loss, eng.ps.grad, eng.ps.flat):
            h.update(t.detach().cpu().numpy().tobytes())
        digests.append(h.hexdigest())
        msg = "run %d sha256 %s" % (run, digests[-1][:16])
        if R is not None:
            rows = []
            for name, gr in R["grads"].items():
                mine, ref = eng.ps.gview(name).detach().cpu().double(), gr.detach().double()
                rows.append((((mine - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item(),
                             ((mine - ref).norm() / ref.norm().clamp_min(1e-30)).item(), name))
            rows.sort(reverse=True)
            msg += "  worst max-norm: " + "; ".join("%s %.2e (L2 %.2e)" % ("/".join(n.split("/")[-4:-2]), m, l) for m, l, n in rows[:3])
        print(msg, flush=True)
        del eng
        torch.cuda.empty_cache()
    print("IDENTICAL" if len(set(digests)) == 1 else "DIFFERENT (%d distinct digests in %d runs)" % (len(set(digests)), a.runs))


if __name__ == "__main__":
    main()
