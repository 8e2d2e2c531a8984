"""Data-parallel exchange for the training step (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The reference has no multi-GPU code; sequences shard over the batch dimension,
so one step needs exactly:
  1. a 1-float all-reduce of t_balance BEFORE the D-gate, so every rank takes the same tf.cond branch
     (reference lib/Teco.py:399,415-417,493-494) -- otherwise the D update diverges across replicas;
  2. one sum all-reduce per optimiser scope over its slice of the flat fp32 gradient buffer (15.3 MB total for
     TecoGAN); the 1/world averaging is folded into the fused Adam kernel's grad_scale.
BatchNorm statistics in D stay per replica (tb = 24 per GPU, as on the single reference GPU).
"""
import torch.distributed as dist


def exchange(grad_flat, scope_ranges, scopes, t_balance=None, group=None):
    """In-place: t_balance <- mean over ranks; grad_flat[a:b] <- SUM over ranks for each scope (caller scales by
    1/world).  Returns the world size."""
    world = dist.get_world_size(group)
    if t_balance is not None:
        dist.all_reduce(t_balance, group=group)
        t_balance.mul_(1.0 / world)
    for scope in scopes:
        a, b = scope_ranges[scope]
        dist.all_reduce(grad_flat[a:b], group=group)
    return world
