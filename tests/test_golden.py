"""The oracle against golden vectors produced by the REFERENCE's own lib/ops.py, lib/frvsr.py, lib/Teco.py
(run unmodified on oracle/tf1_shim.py by oracle/make_golden.py; see those files for what this does and does not pin)."""
import os

import numpy as np
import pytest
import torch

from oracle import make_golden as MG
from oracle import ops as O
from oracle import teco as OT

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_in_tree_numerics_match_reference_ops():
    g = np.load(os.path.join(GOLD, "reference_ops.npz"))
    x = torch.from_numpy(g["x"])
    assert np.array_equal(O.upscale_four(x).numpy(), g["upscale_four"])          # bit-exact restatement
    assert np.array_equal(O.bicubic_four(x).numpy(), g["bicubic_four"])
    assert np.array_equal(O.preprocess(x).numpy(), g["preprocess"]) and np.array_equal(O.deprocess(x).numpy(), g["deprocess"])
    import lib.ops as L                                                          # the product's helper, CPU-only function
    assert np.allclose(L.gaussian_2dkernel(9, 1.5), g["gauss9"], atol=1e-15)


@pytest.mark.parametrize("name", sorted(MG.CASES))
def test_training_step_matches_reference_wiring(name):
    g = np.load(os.path.join(GOLD, "reference_wiring.npz"))
    kw, gan = MG.CASES[name]
    F = OT.default_flags(**kw)
    st = OT.State(F, seed=42, gan=gan)
    x, y = MG.batch(F)
    R = OT.train_step(st, x, y)
    B, T = F.batch_size, R["gen_outputs"].shape[1]
    gen = R["gen_outputs"].reshape(B * T, *R["gen_outputs"].shape[2:])          # (b, t) order = s_gen_output
    assert np.allclose(gen[:, ::8, ::8].detach().numpy(), g[name + "/gen_out_slice"], atol=1e-6)
    assert np.allclose(MG.fingerprint(gen), g[name + "/gen_out_fp"], rtol=1e-6, atol=1e-6)
    mine = dict(zip(R["names"], [float(v) for v in R["vals"]]))
    for n_, v in zip(g[name + "/loss_names"], g[name + "/losses"]):              # update_list_name / update_list
        assert str(n_) in mine, n_
        assert abs(mine[str(n_)] - v) <= 1e-5 * max(1.0, abs(v)), (n_, mine[str(n_)], v)
    names = [str(s) for s in g[name + "/var_names"]]
    assert names == sorted(R["grads"])                                           # same trainable variable set/names
    for i, k in enumerate(names):
        ref = g[name + "/grad_fp"][i]
        got = MG.fingerprint(R["grads"][k])
        assert np.allclose(got, ref, rtol=1e-4, atol=1e-5 * max(abs(ref[1]), 1e-3)), ("grad", k)
        refw = g[name + "/weight_fp"][i]
        gotw = MG.fingerprint(st.P[k])
        assert np.allclose(gotw, refw, rtol=1e-4, atol=3 * F.learning_rate * np.sqrt(refw[2])), ("weight", k)
    assert st.global_step == int(g[name + "/global_step"]) == 1


@pytest.mark.skipif(not os.path.isdir("/root/reference/lib"), reason="reference sources only exist in the build container")
def test_goldens_regenerate_from_the_reference_sources():
    MG.main(check_only=True)             # re-runs the reference on the shim and asserts reference == oracle
