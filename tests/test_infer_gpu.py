"""Inference recurrence (reference main.py:195-260) on the HIP path vs the CPU oracle."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import nets as ON
from oracle import teco as OT
from tecogan_amd.infer import InferenceEngine
from tecogan_amd.params import damp_values
from util import assert_close_per_elem

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def params(nres, damp=False):
    P = ON.init_params(ON.generator_spec(nres), 42)
    P.update(ON.init_params(ON.fnet_spec(), 43))
    return damp_values(P) if damp else P


def run(h, w, nres, frames, act_dtype, use_graph):
    P = params(nres)
    g = torch.Generator().manual_seed(5)
    seq = [torch.rand(1, h, w, 3, generator=g) for _ in range(frames)]
    st = OT.InferenceState(h, w)
    eng = InferenceEngine(nres, h, w, "cuda", act_dtype, use_graph=use_graph)
    eng.load(P)
    errs = []
    for f in seq:
        ref = OT.inference_step(P, st, f, nres)
        out = eng.step(f.cuda()).cpu()
        errs.append(((out - ref).abs().max() / ref.abs().max()).item())
    return errs


@pytest.mark.parametrize("h,w", [(16, 24), (18, 20), (36, 45)])
def test_inference_fp32_parity(h, w):
    """Includes sizes that are not multiples of 8 (calendar is 144x180: ow=4; 270x480: oh=6)."""
    errs = run(h, w, 2, 4, torch.float32, use_graph=False)
    assert max(errs) < 1e-3, errs


def test_inference_graph_replay_matches():
    errs = run(18, 20, 2, 5, torch.float32, use_graph=True)
    assert max(errs) < 1e-3, errs


def test_inference_bf16_bounded():
    errs = run(16, 24, 2, 4, torch.bfloat16, use_graph=True)
    assert max(errs) < 3e-2, errs


def stream_parity(seq, h, w, nres, tag, tol=1e-3):
    """fp32 HIP stream vs the oracle on EVERY frame, per pixel: |a-b| <= tol * max(|b|, 1e-3 max|b|).  Damped xavier
    weights (params.damp_values): the regime of a trained generator, where the recurrence is well conditioned."""
    P = params(nres, damp=True)
    st = OT.InferenceState(h, w)
    eng = InferenceEngine(nres, h, w, "cuda", torch.float32, use_graph=True)
    eng.load(P)
    worst = 0.0
    for i, f in enumerate(seq):
        ref = OT.inference_step(P, st, f, nres)
        out = eng.step(f.cuda()).cpu()
        worst = max(worst, assert_close_per_elem(out, ref, tol, 1e-3, what="%s frame %d" % (tag, i)))
        assert ref.min().item() > -0.5 and ref.max().item() < 1.5, "recurrence left the image range: ill-conditioned test"
    print("\n[%s] %d frames, worst per-pixel relative error %.2e" % (tag, len(seq), worst))


def test_calendar_clip_fp32_parity_every_frame():
    """BASELINE configs[0] / SURVEY 8c.10: the reference's own LR/calendar clip through the loop of main.py:253-260 --
    41 PNGs + the 5 mirrored warm-up frames of lib/dataloader.py:42-44 = 46 frames of 144x180 (ow = 4: the FNet output
    is 144x176 and is SYMMETRIC-padded, main.py:188-190,212), num_resblock=16."""
    z = np.load(os.path.join(GOLD, "calendar_lr.npz"))
    frames = z["frames"]
    assert frames.shape == (41, 144, 180, 3) and frames.dtype == np.uint8
    png0 = str(z["png0_sha256"])
    assert png0.startswith("0be6a70a") and png0.endswith("754c35")                     # sha256 of LR/calendar/0001.png
    assert hashlib.sha256(frames[0].tobytes()).hexdigest() == str(z["rgb0_sha256"])     # decoded RGB pixels
    seq = [torch.from_numpy(f.astype(np.float32) / 255.0)[None] for f in frames]
    seq = seq[5:0:-1] + seq
    assert len(seq) == 46
    stream_parity(seq, 144, 180, 16, "calendar")


def test_inference_270x480_fp32_parity():
    """BASELINE configs[4] geometry (480x270 -> 1920x1080, oh = 6), 8 frames of a smooth synthetic clip."""
    g = torch.Generator().manual_seed(9)
    base = torch.nn.functional.interpolate(torch.rand(1, 3, 18, 32, generator=g), size=(270 + 16, 480 + 16), mode="bicubic",
                                           align_corners=False).clamp(0, 1).permute(0, 2, 3, 1)
    seq = [base[:, 2 * i:2 * i + 270, i:i + 480].contiguous() for i in range(8)]        # a slow pan
    stream_parity(seq, 270, 480, 16, "270x480")
