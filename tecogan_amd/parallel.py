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

from . import kernels as K


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


class ExchangeMixin:
    """The gradient exchange of TrainEngine: with the RCCL backend the collectives are captured segments on the
    communication stream "C" -- `ar_d` (balance scalar + D gradients, as soon as D's own-gradient passes are done: overlaps the
    BPTT), `ar_g` (generator, after its weight gradients: overlaps FNet's backward pass), `ar_f` (FNet) -- and `update` joins
    them; backends that cannot be captured (gloo) run `exchange()` above as one eager segment between the compute and update
    graphs.  Expects: self.exchange_mode, self.world, self.standin, self.pg, self.ps, self.opt_scopes, self.gan, self._slot(),
    the segment machinery of segments.SegmentRunner."""

    def _exchange_seg(self, name, scopes, after, with_balance=False):
        """captured mode: all-reduce `scopes` of the flat gradient buffer as a segment of the communication stream,
        ordered after the segments `after`; it overlaps whatever the compute streams do next, `update` joins."""
        if self.exchange_mode != "captured" or self._skip_update:
            # eval_losses (validation on ONE rank, main.py) must not issue collectives: an all-reduce from rank 0 alone would
            # pair with the other ranks' next training step and shift every later collective by one
            return
        with self._seg(name, "C", after):
            if with_balance and self.gan:            # every rank must take the same D-gate branch (lib/Teco.py:493-494)
                tb = self._slot("t_balance")
                self._sum_all_reduce(tb)
                K.affine(tb, tb, 1.0 / self.world, 0.0)
            for scope in scopes:
                a, b = self.ps.scope_range[scope]
                self._sum_all_reduce(self.ps.grad[a:b])
        self.exchange_segments.append(name)

    def _sum_all_reduce(self, t):
        if self.standin > 1:                         # test stand-in: W identical ranks
            K.affine(t, t, float(self.standin), 0.0)
        else:
            dist.all_reduce(t, group=self.pg)

    def allreduce_bytes(self):
        """Bytes every rank contributes to the gradient exchange of one step (fp32 flat buffers + the balance scalar)."""
        n = sum(self.ps.scope_range[s][1] - self.ps.scope_range[s][0] for s in self.opt_scopes)
        return 4 * n + (4 if self.gan else 0)

    def _allreduce(self):
        tbv = self._slot("t_balance") if self.gan else None
        exchange(self.ps.grad, self.ps.scope_range, self.opt_scopes, tbv, self.pg)
