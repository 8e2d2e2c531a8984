#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own lib/ops.py, lib/frvsr.py, lib/Teco.py
(imported unmodified from /root/reference on top of oracle/tf1_shim.py) on seeded inputs and weights.

    python -m oracle.make_golden            # writes tests/golden/reference_wiring.npz, reference_ops.npz

What this pins: the reference's graph wiring and in-tree numerics (see oracle/tf1_shim.py docstring).  What it does not:
TensorFlow's own op numerics (supplied by oracle/ops.py).  /root/reference only exists in the build container, so the
vectors are committed and the script is kept for provenance; `tests/test_golden.py` re-derives them with the oracle.
Large tensors are stored as fingerprints (sum, L2 norm, 24 strided samples) to keep the fixtures small.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import nets as ON  # noqa: E402
from oracle import teco as OT  # noqa: E402
from oracle import tf1_shim as S  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

CASES = {
    # name: (flags kwargs, GAN?)
    "tecogan": (dict(batch_size=2, RNN_N=3, crop_size=16, num_resblock=2), True),
    "frvsr": (dict(batch_size=2, RNN_N=3, crop_size=16, num_resblock=2, pingpang=False, ratio=-0.01, vgg_scaling=-0.2), False),
    "tecogan_nopp": (dict(batch_size=2, RNN_N=3, crop_size=16, num_resblock=1, pingpang=False, vgg_scaling=-0.2), True),
}


def fingerprint(t):
    t = t.detach().double().reshape(-1)
    n = t.numel()
    idx = torch.linspace(0, n - 1, 24).long()            # 24 strided samples (repeats for tiny tensors)
    return np.concatenate(([t.sum().item(), t.norm().item(), float(n)], t[idx].numpy())).astype(np.float64)


def batch(F, seed=1234):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(F.batch_size, F.RNN_N, F.crop_size, F.crop_size, 3, generator=g)
    y = torch.rand(F.batch_size, F.RNN_N, 4 * F.crop_size, 4 * F.crop_size, 3, generator=g) * 2 - 1
    return x, y


def ref_flags(F):
    """The attributes the reference's TecoGAN() reads from FLAGS (lib/Teco.py)."""
    return SimpleNamespace(**vars(F))


def run_reference(name):
    kw, gan = CASES[name]
    F = OT.default_flags(**kw)
    st = OT.State(F, seed=42, gan=gan)
    preset = dict(st.P)
    if st.vgg is not None:
        preset.update(st.vgg)
    preset.update(st.bn)
    S.reset(preset)
    x, y = batch(F)
    with S.reference_modules() as RT:
        net = RT.TecoGAN(S.T(x), S.T(y), ref_flags(F), gan)
        names = list(net.update_list_name)
        losses = [float(S._u(v)) for v in net.update_list]
        gen_out = S._u(net.gen_output).detach().clone()
        grads = dict(S._STATE.captured_grads)
        S.flush()                                        # D (gate permitting), G, fnet Adam steps; EMA updates
        weights = {k: v.v.detach().clone() for k, v in S._STATE.vars.items() if k in st.P}
        step = float(S._STATE.vars["global_step"].v)
    return F, gan, st, (x, y), dict(names=names, losses=losses, gen_out=gen_out, grads=grads, weights=weights, step=step)


def run_reference_ops():
    """In-tree numerics straight from the reference's lib/ops.py."""
    S.reset({})
    g = torch.Generator().manual_seed(7)
    x = torch.rand(2, 5, 6, 3, generator=g) * 2 - 1
    with S.reference_modules() as RT:
        import lib.ops as RO
        out = dict(x=x.numpy(), upscale_four=S._u(RO.upscale_four(S.T(x))).numpy(),
                   bicubic_four=S._u(RO.bicubic_four(S.T(x))).numpy(),
                   preprocess=S._u(RO.preprocess(S.T(x))).numpy(), deprocess=S._u(RO.deprocess(S.T(x))).numpy(),
                   gauss9=RO.gaussian_2dkernel(9, 1.5).astype(np.float64))
    return out


def main(check_only=False):
    os.makedirs(GOLD, exist_ok=True)
    rec = {}
    for name in CASES:
        F, gan, st, (x, y), ref = run_reference(name)
        # cross-check against the oracle restatement before writing anything
        st2 = OT.State(F, seed=42, gan=gan)
        R = OT.train_step(st2, x, y)
        B, T = F.batch_size, R["gen_outputs"].shape[1]
        o_gen = R["gen_outputs"].reshape(B * T, *R["gen_outputs"].shape[2:])
        err = (o_gen - ref["gen_out"]).abs().max().item()
        assert err < 1e-5, (name, "gen_output", err)
        o_l = dict(zip(R["names"], [float(v) for v in R["vals"]]))
        for n_, v in zip(ref["names"], ref["losses"]):
            if n_ in o_l:
                assert abs(o_l[n_] - v) < 1e-5 * max(1, abs(v)), (name, n_, o_l[n_], v)
        for k, g_ in ref["grads"].items():
            e = (R["grads"][k] - g_).abs().max().item() / max(g_.abs().max().item(), 1e-12)
            assert e < 1e-4, (name, "grad", k, e)
        for k, w_ in ref["weights"].items():
            e = (st2.P[k] - w_).abs().max().item()
            assert e < 1e-6 + 2.1 * F.learning_rate * (e > 1e-6), (name, "weight", k, e)
        print("%-14s reference == oracle: gen %.1e, %d losses, %d grads, %d weights" %
              (name, err, len(ref["losses"]), len(ref["grads"]), len(ref["weights"])))
        rec[name + "/loss_names"] = np.array(ref["names"])
        rec[name + "/losses"] = np.array(ref["losses"], dtype=np.float64)
        rec[name + "/gen_out_fp"] = fingerprint(ref["gen_out"])
        rec[name + "/gen_out_slice"] = ref["gen_out"][:, ::8, ::8].numpy()
        rec[name + "/var_names"] = np.array(sorted(ref["grads"]))
        rec[name + "/grad_fp"] = np.stack([fingerprint(ref["grads"][k]) for k in sorted(ref["grads"])])
        rec[name + "/weight_fp"] = np.stack([fingerprint(ref["weights"][k]) for k in sorted(ref["grads"])])
        rec[name + "/global_step"] = np.array(ref["step"])
    ops = run_reference_ops()
    if not check_only:
        np.savez_compressed(os.path.join(GOLD, "reference_wiring.npz"), **rec)
        np.savez_compressed(os.path.join(GOLD, "reference_ops.npz"), **ops)
        print("wrote", GOLD)


def calendar_fixture(check_only=False):
    """The reference's only real-data fixture: LR/calendar/0001..0041.png (runGan.py case 1, main.py:185-270), decoded
    to RGB uint8 exactly as the loader does (cv.imread(...)[:, :, ::-1] of lib/dataloader.py:32 == PIL RGB).  The GPU box
    has no /root/reference, so the decoded clip is committed (tests/golden/calendar_lr.npz, 41 x 144 x 180 x 3)."""
    import hashlib
    from PIL import Image
    d = "/root/reference/LR/calendar"
    names = sorted(f for f in os.listdir(d) if f.endswith(".png"))
    names.sort(key=lambda f: int("".join(ch for ch in f if ch.isdigit()) or -1))
    frames = np.stack([np.asarray(Image.open(os.path.join(d, n)).convert("RGB")) for n in names])
    png0 = hashlib.sha256(open(os.path.join(d, names[0]), "rb").read()).hexdigest()
    rgb0 = hashlib.sha256(frames[0].tobytes()).hexdigest()
    assert frames.shape == (41, 144, 180, 3) and frames.dtype == np.uint8
    assert png0.startswith("0be6a70a") and png0.endswith("754c35"), png0
    out = os.path.join(GOLD, "calendar_lr.npz")
    if check_only:
        got = np.load(out)
        assert np.array_equal(got["frames"], frames) and str(got["png0_sha256"]) == png0
        print("calendar fixture matches /root/reference/LR/calendar")
        return
    np.savez_compressed(out, frames=frames, names=np.array(names), png0_sha256=np.array(png0), rgb0_sha256=np.array(rgb0))
    print("wrote", out, frames.shape)


if __name__ == "__main__":
    main(check_only="--check" in sys.argv)
    calendar_fixture(check_only="--check" in sys.argv)
