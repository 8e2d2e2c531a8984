"""Drop-in for the reference's `lib/frvsr.py`: `fnet` and `generator_F` with the reference signatures,
executed by the fused HIP kernel schedules of `tecogan_amd.nets` (not op by op).

Variables are created under the caller's variable scope with the reference's names
(`<scope>/autoencode_unit/...`, `<scope>/generator_unit/...`) and are views into one
flat parameter buffer, so `lib.ops.global_variables()` / checkpoints see ordinary named tensors.
"""
from collections import OrderedDict

import torch

from lib import ops as _ops
from lib.ops import *  # noqa: F401,F403  (the reference's lib.frvsr re-exports lib.ops via lib.dataloader)
from tecogan_amd import kernels as K
from tecogan_amd.nets import FNET_CPAD, GEN_CPAD, FNet, Generator
from tecogan_amd.params import ParamStore, fnet_spec, generator_spec, init_values

_NETS = {}        # (scope path, unit) -> (ParamStore, net)


def _scope_path():
    return "/".join(s for s, _ in _ops._SCOPE)


def _get_net(unit, default_root, spec, make, reuse, device):
    """Create (reuse=False) or fetch (reuse=True) the parameter store of one network under the current scope."""
    root = _scope_path() or default_root
    key = (root, unit)
    if key in _NETS:
        if not reuse:
            raise ValueError("Variable %s/%s/... already exists, disallowed. Did you mean to set reuse=True?" % key)
        return _NETS[key]
    if reuse:
        raise ValueError("Variable %s/%s/... does not exist, or was not created with reuse=False" % key)
    ps = ParamStore(OrderedDict([(default_root, spec)]), device, torch.float32, trainable=False)
    ps.load(init_values(spec, _ops._SEED[0] + len(_NETS)))
    for name in ps.entries:                              # publish under the caller's scope, as views
        _ops._VARS[root + name[len(default_root):]] = ps.view(name)
    _NETS[key] = (ps, make(ps))
    return _NETS[key]


def sync_variables():
    """Call after assigning into lib.ops.global_variables() tensors (checkpoint restore): refreshes the
    MFMA weight copies of every network built through this module."""
    for ps, _ in _NETS.values():
        ps.repack()


def fnet(fnet_input, reuse=False):
    """Flow estimator (reference lib/frvsr.py:4-41).  fnet_input [N,h,w,6] = concat(prev LR, cur LR);
    returns the LR flow [N,h',w',2] in LR pixels (|flow| <= 24), h' = h - h%8."""
    x = _ops._need_cuda(fnet_input)
    if x.shape[-1] != 6:
        raise ValueError("fnet expects 6 input channels (two RGB frames), got %d" % x.shape[-1])
    ps, net = _get_net("autoencode_unit", "fnet", fnet_spec(), FNet, reuse, x.device)
    xin = K.concat2_pad(x, None, torch.empty(*x.shape[:-1], FNET_CPAD, device=x.device))
    flow, _ = net.forward(xin, keep=False)
    return flow


def generator_F(gen_inputs, gen_output_channels, reuse=False, FLAGS=None):
    """Recurrent SR generator (reference lib/frvsr.py:44-88).  gen_inputs [N,h,w,51] = concat(LR frame,
    space_to_depth(warped previous HR, 4)); returns the HR frame [N,4h,4w,3] in [-1,1]."""
    if FLAGS is None:
        raise ValueError('No FLAGS is provided for generator')
    x = _ops._need_cuda(gen_inputs)
    if x.shape[-1] != 51 or gen_output_channels != 3:
        raise ValueError("generator_F: the HIP path is built for 3+48 input and 3 output channels")
    nres = FLAGS.num_resblock
    ps, net = _get_net("generator_unit", "generator", generator_spec(nres), lambda p: Generator(p, nres), reuse, x.device)
    xin = K.concat2_pad(x, None, torch.empty(*x.shape[:-1], GEN_CPAD, device=x.device))
    out, _ = net.forward(xin, keep=False)
    return out
