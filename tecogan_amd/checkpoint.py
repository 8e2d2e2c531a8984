"""Checkpoint I/O for the reference's variable names.

Two on-disk forms, both keyed by the TF variable names:
  * a torch file `model-<step>` = {"variables", "adam_m", "adam_v", "sched", "global_step"} (exact resume of this backend);
  * a TensorFlow tensor bundle `model-<step>.index` / `.data-00000-of-00001` (tecogan_amd/tf_bundle.py), the format the
    reference's `tf.train.Saver` reads and writes (reference main.py:224,245,307-352,365,420).  Variables are stored under
    their TF names, Adam slots under TF's slot names (`<scope>/<variable>/Adam`, `/Adam_1`) plus `global_step`.  The scope is
    `generator_train` for ALL THREE optimisers: slots are created by `apply_gradients`, and the reference calls it for the
    discriminator too inside `with tf.variable_scope('generator_train')` (lib/Teco.py:438,463 -- the optimiser OBJECT is built
    under 'tdicriminator_train' [sic], :420, but that scope owns no variable).  Interoperability status: the VARIABLE names
    are TF's (weights-only restore of `model/TecoGAN`-style files); the optimiser-state names are read off the reference graph
    and have never been checked against a TF-written training checkpoint (none exists offline).
`load_variables(path)` accepts the path of a torch file, a bundle prefix (`model/TecoGAN`-style prefixes of the pre-trained
models) or a TensorFlow V1 tensor-slice file (slim's original `vgg_19.ckpt`).
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from . import tf_bundle

OPT_SCOPE = {"generator": "generator_train", "fnet": "generator_train", "tdiscriminator": "generator_train"}
LEGACY_SCOPES = ("generator_train", "tdicriminator_train")           # round-2 files wrote D's slots under the latter


def power_keys(scopes):
    """TF names of each optimiser's bias-correction accumulators.  They are non-slot variables created by the first
    `apply_gradients` of each optimiser, all under variable_scope('generator_train') and uniquified in creation order:
    GAN graph (lib/Teco.py:463-468): discriminator, generator, fnet -> beta1_power, beta1_power_1, beta1_power_2;
    FRVSR graph (lib/Teco.py:446-449): generator, fnet -> beta1_power, beta1_power_1."""
    order = [s for s in ("tdiscriminator", "generator", "fnet") if s in scopes]
    return {s: "generator_train/beta1_power" + ("_%d" % i if i else "") for i, s in enumerate(order)}


STEPS_KEY = "tecogan_amd/adam_steps/"          # this backend's exact integer Adam step counts (one int64 per optimiser scope)
TB_EMA_KEY = "tecogan_amd/t_balance_ema"      # this backend's key for the EMA(0.99) shadow of t_balance (Teco.py:415-417)


def _is_torch_file(path):
    return os.path.isfile(path) and not tf_bundle.is_v1_checkpoint(path)


def load_variables(path):
    """-> (variables: OrderedDict name -> torch tensor, extra: dict).  `extra` holds what a full resume needs when the
    file has it: "adam_m"/"adam_v" (flat torch buffers of the torch form, or name -> tensor dicts of a bundle),
    "sched", "global_step"."""
    if _is_torch_file(path):
        ck = torch.load(path, map_location="cpu")
        return OrderedDict(ck["variables"]), {k: v for k, v in ck.items() if k != "variables"}
    if tf_bundle.is_v1_checkpoint(path):                                # TF V1 tensor-slice file (slim's vgg_19.ckpt)
        vals = tf_bundle.read_v1_checkpoint(path)
        return OrderedDict((k, torch.from_numpy(v.astype(np.float32))) for k, v in vals.items()
                           if v.dtype in (np.float32, np.float64, np.float16)), {}
    if not tf_bundle.is_bundle(path):
        raise ValueError("checkpoint %s not found (neither a torch file nor a TensorFlow bundle prefix)" % path)
    r = tf_bundle.BundleReader(path)
    variables, m, v, extra = OrderedDict(), {}, {}, {}
    import math
    keys = set(r.keys())
    b1 = float(r.get(TB_EMA_KEY + "/beta1")) if (TB_EMA_KEY + "/beta1") in keys else 0.9
    if TB_EMA_KEY in keys:
        extra["tb_ema"] = float(r.get(TB_EMA_KEY))
    scopes_here = [sc for sc in ("tdiscriminator", "generator", "fnet") if any(k.startswith(sc + "/") for k in keys)]
    steps = {}
    pmap = power_keys(scopes_here)
    if "tdicriminator_train/beta1_power" in keys:
        # round-2 layout of this backend: D's accumulators under its own (misspelt, as in the reference) scope, the generator's
        # and FNet's as beta1_power / beta1_power_1 -- detected FIRST: read with today's mapping the generator's count would
        # land on D (whose count lags under the gate) and FNet's on the generator
        pmap = {"tdiscriminator": "tdicriminator_train/beta1_power", "generator": "generator_train/beta1_power",
                "fnet": "generator_train/beta1_power_1"}
        pmap = {sc: k for sc, k in pmap.items() if sc in scopes_here}
    for scope, pk in pmap.items():
        if STEPS_KEY + scope in keys:                      # exact count written by this backend
            steps[scope] = int(r.get(STEPS_KEY + scope))
            continue
        # TF stores beta^(t+1) after t updates.  beta2_power = 0.999^(t+1) stays a normal float32 up to t ~ 8e4, while
        # beta1_power = 0.9^(t+1) goes denormal near t ~ 830 and underflows at ~980: prefer beta2_power.
        pk2 = pk.replace("beta1_power", "beta2_power")
        if pk2 in keys and 0.0 < float(r.get(pk2)) < 1.0:
            steps[scope] = max(int(round(math.log(float(r.get(pk2))) / math.log(0.999))) - 1, 0)
        elif pk in keys and 0.0 < float(r.get(pk)) < 1.0:
            steps[scope] = max(int(round(math.log(float(r.get(pk))) / math.log(b1))) - 1, 0)
    if steps:
        extra["adam_steps"] = steps
    for key in r.keys():
        if key.startswith(TB_EMA_KEY):
            continue
        if key.endswith("/Adam") or key.endswith("/Adam_1"):
            base = key.rsplit("/", 1)[0]
            for scope in LEGACY_SCOPES:                               # strip the optimizer's variable scope
                if base.startswith(scope + "/"):
                    base = base[len(scope) + 1:]
            (m if key.endswith("/Adam") else v)[base] = torch.from_numpy(r.get(key))
            continue
        if key == "global_step":
            extra["global_step"] = int(r.get(key))
            continue
        if "beta1_power" in key or "beta2_power" in key or "ExponentialMovingAverage" in key:
            continue
        a = r.get(key)
        if a.dtype in (np.float32, np.float64, np.float16):
            variables[key] = torch.from_numpy(a.astype(np.float32))
    if m and v:
        extra["adam_m"], extra["adam_v"] = m, v
    return variables, extra


def save_bundle(prefix, ps, global_step, beta1=0.9, beta2=0.999, adam_steps=None, tb_ema=None):
    """Write the parameter store (variables + Adam slots + global_step) as a TensorFlow tensor bundle.
    adam_steps: scope -> number of Adam updates applied (the gated discriminator lags global_step); tb_ema: the EMA shadow
    of t_balance.  Interoperability note: variables (inference / pre_trained_model restore) follow TF's names exactly; the
    optimiser-state names follow the reference graph as far as it can be read offline (no TF-written file to check against)."""
    out = OrderedDict()
    for name, e in ps.entries.items():
        out[name] = ps.view(name).detach().cpu().numpy()
        scope = OPT_SCOPE.get(e["scope"])
        if scope is not None and ps.trainable:
            out["%s/%s/Adam" % (scope, name)] = ps.view(name, ps.m).detach().cpu().numpy()
            out["%s/%s/Adam_1" % (scope, name)] = ps.view(name, ps.v).detach().cpu().numpy()
    out["global_step"] = np.asarray(int(global_step), dtype=np.int64)
    scopes = [sc for sc in ("tdiscriminator", "generator", "fnet") if any(e["scope"] == sc for e in ps.entries.values())]
    for scope, pk in power_keys(scopes).items():
        t = max(int((adam_steps or {}).get(scope, global_step)), 0)
        out[pk] = np.asarray(beta1 ** (t + 1), dtype=np.float32)                        # TF stores beta^(t+1) after t updates
        out[pk.replace("beta1_power", "beta2_power")] = np.asarray(beta2 ** (t + 1), dtype=np.float32)
        out[STEPS_KEY + scope] = np.asarray(t, dtype=np.int64)
    out[TB_EMA_KEY + "/beta1"] = np.asarray(beta1, dtype=np.float32)
    if tb_ema is not None:
        out[TB_EMA_KEY] = np.asarray(tb_ema, dtype=np.float32)
    return tf_bundle.write_bundle(prefix, out)
