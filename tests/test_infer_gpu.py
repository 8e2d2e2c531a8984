"""Inference recurrence (reference main.py:195-260) on the HIP path vs the CPU oracle."""
import pytest
import torch

from oracle import nets as ON
from oracle import teco as OT
from tecogan_amd.infer import InferenceEngine

pytestmark = pytest.mark.gpu


def run(h, w, nres, frames, act_dtype, use_graph):
    P = ON.init_params(ON.generator_spec(nres), 42)
    P.update(ON.init_params(ON.fnet_spec(), 43))
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
