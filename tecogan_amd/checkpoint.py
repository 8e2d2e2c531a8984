"""Checkpoint I/O for the reference's variable names.

Two on-disk forms, both keyed by the TF variable names:
  * a torch file `model-<step>` = {"variables", "adam_m", "adam_v", "sched", "global_step"} (exact resume of this backend);
  * a TensorFlow tensor bundle `model-<step>.index` / `.data-00000-of-00001` (tecogan_amd/tf_bundle.py), the format the
    reference's `tf.train.Saver` reads and writes (reference main.py:224,245,307-352,365,420).  Variables are stored under
    their TF names, Adam slots under TF's slot names (`<optimizer scope>/<variable>/Adam`, `/Adam_1`; optimizer scopes
    `generator_train`, `tdicriminator_train` [sic], reference lib/Teco.py:420,439), plus `global_step`.
`load_variables(path)` accepts the path of a torch file, a bundle prefix (`model/TecoGAN`-style prefixes of the pre-trained
models) or a TensorFlow V1 tensor-slice file (slim's original `vgg_19.ckpt`).
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from . import tf_bundle

OPT_SCOPE = {"generator": "generator_train", "fnet": "generator_train", "tdiscriminator": "tdicriminator_train"}
# TF names of each optimiser's bias-correction accumulators: the generator and FNet AdamOptimizers are both created under
# variable_scope('generator_train') (lib/Teco.py:438-440), so the second one gets the "_1" suffix.
POWER_KEY = {"generator": "generator_train/beta1_power", "fnet": "generator_train/beta1_power_1",
             "tdiscriminator": "tdicriminator_train/beta1_power"}
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
    steps = {}
    for key in r.keys():
        if key == TB_EMA_KEY:
            extra["tb_ema"] = float(r.get(key))
            continue
        for scope, pk in POWER_KEY.items():
            if key == pk:                    # TF stores beta1^(t+1) after t updates
                p1 = float(r.get(key))
                b1 = float(r.get(TB_EMA_KEY + "/beta1")) if (TB_EMA_KEY + "/beta1") in r.keys() else 0.9
                if 0.0 < p1 < 1.0:
                    steps[scope] = max(int(round(math.log(p1) / math.log(b1))) - 1, 0)
    if steps:
        extra["adam_steps"] = steps
    for key in r.keys():
        if key.startswith(TB_EMA_KEY):
            continue
        if key.endswith("/Adam") or key.endswith("/Adam_1"):
            base = key.rsplit("/", 1)[0]
            for scope in set(OPT_SCOPE.values()):                     # strip the optimizer's variable scope
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
    for scope in sorted(set(e["scope"] for e in ps.entries.values() if e["scope"] in POWER_KEY)):
        t = max(int((adam_steps or {}).get(scope, global_step)), 0)
        out[POWER_KEY[scope]] = np.asarray(beta1 ** (t + 1), dtype=np.float32)         # TF stores beta^(t+1) after t updates
        out[POWER_KEY[scope].replace("beta1_power", "beta2_power")] = np.asarray(beta2 ** (t + 1), dtype=np.float32)
    out[TB_EMA_KEY + "/beta1"] = np.asarray(beta1, dtype=np.float32)
    if tb_ema is not None:
        out[TB_EMA_KEY] = np.asarray(tb_ema, dtype=np.float32)
    return tf_bundle.write_bundle(prefix, out)
