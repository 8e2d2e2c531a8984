"""End-to-end parity of the HIP training step against the CPU oracle on identical seeded inputs/weights.

Tolerance (fp32 mode): 1e-3 relative per tensor (north-star bar), in practice ~1e-5.  bf16 mode reports
its own measured error against the fp32 oracle with a looser, stated bound."""
import os

import pytest
import torch

from oracle import teco as OT
from tecogan_amd.engine import TrainEngine

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def make_batch(B, T, cs, seed=1234):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, T, cs, cs, 3, generator=g)
    y = torch.rand(B, T, 4 * cs, 4 * cs, 3, generator=g) * 2 - 1
    return x, y


def frame_major(gen_outputs):          # oracle [B,T,...] -> engine [T,B,...]
    return gen_outputs.transpose(0, 1)


def run_pair(F, gan, steps=1, act_dtype=torch.float32, use_graph=False, damp=False, oracle_dtype=torch.float32):
    S = OT.State(F, seed=42, gan=gan, dtype=oracle_dtype)
    if damp:        # a well-conditioned recurrence (tecogan_amd.params.damp_values): see check_full_config
        from tecogan_amd.params import damp_values
        S.P = damp_values(S.P)
    eng = TrainEngine(F, DEV, gan=gan, act_dtype=act_dtype, seed=7, use_graph=use_graph)
    eng.ps.load(S.P)
    if eng.use_vgg:
        eng.vps.load(S.vgg)
    x, y = make_batch(F.batch_size, F.RNN_N, F.crop_size)
    out = []
    for _ in range(steps):
        R = OT.train_step(S, x.to(oracle_dtype), y.to(oracle_dtype))
        eng.step(x.to(DEV), y.to(DEV))
        torch.cuda.synchronize()
        out.append(R)
    return S, eng, out


def check_step(S, eng, R, tol, adam_slack=0.02):
    assert rel_err(eng.gen, frame_major(R["gen_outputs"])) < tol, "gen_outputs"
    L = eng.losses()
    for name, val in zip(R["names"], R["vals"]):
        if name in L:
            assert abs(L[name] - float(val)) <= tol * max(1.0, abs(float(val))), (name, L[name], float(val))
    # Gradients: relative L2 error <= tol per tensor, and max-abs error <= 10*tol of the tensor's max.  (In these
    # deliberately tiny configurations a single ReLU pre-activation within fp32 rounding of 0 can flip its 0/1
    # mask between summation orders; that moves a 9x64 slice of one weight gradient by ~1e-3 of its max while the
    # tensor as a whole still agrees to ~1e-5 -- max-abs alone made the test flaky.)
    for name, g in R["grads"].items():
        mine = eng.ps.gview(name).detach().cpu().double()
        ref = g.detach().double()
        l2 = ((mine - ref).norm() / ref.norm().clamp_min(1e-30)).item()
        assert l2 < tol, "gradient %s relative L2 error %g" % (name, l2)
        assert rel_err(mine, ref) < 10 * tol, "gradient %s max-abs rel err %g" % (name, rel_err(mine, ref))
    # Adam's update lr*m/(sqrt(v)+eps) is ill-conditioned where |g| ~ eps: a gradient element that is
    # mathematically ~0 comes out as +-1e-8 rounding noise and its first-step update flips between +-lr.
    # Single step: elements whose oracle gradient is well above the gradient noise floor must match to `tol`
    # (+2% of an lr step); the others may differ by a sign flip (2 lr).  Multi-step runs (adam_slack=2.0) allow
    # 2 lr per step everywhere -- the gradients themselves are already held to `tol` above.
    lr = S.flags.learning_rate
    steps = max(S.global_step, 1)
    for name, p in S.P.items():
        d = (eng.ps.view(name).detach().cpu() - p).abs()
        bound = torch.full_like(p, tol * p.abs().max().item() + adam_slack * lr * steps)
        if steps == 1 and name in R["grads"]:
            g = R["grads"][name]
            gerr = (eng.ps.gview(name).detach().cpu() - g).abs().max().item()
            noisy = g.abs() < max(100.0 * gerr, 1e-6)
            bound = torch.where(noisy, torch.full_like(p, 2.0 * lr) + bound, bound)
        assert bool((d <= bound).all()), "post-Adam weight %s: max |diff| %g" % (name, d.max().item())


def test_frvsr_step_fp32_parity():
    F = OT.frvsr_flags(batch_size=2, RNN_N=3, crop_size=16, num_resblock=2)
    S, eng, Rs = run_pair(F, gan=False)
    check_step(S, eng, Rs[-1], 1e-3)


def test_frvsr_two_steps_graph_replay():
    """Same program captured in a hipGraph and replayed twice == oracle after two steps."""
    F = OT.frvsr_flags(batch_size=2, RNN_N=3, crop_size=16, num_resblock=2)
    S, eng, Rs = run_pair(F, gan=False, steps=2, use_graph=True)
    check_step(S, eng, Rs[-1], 1e-3)
    assert eng.global_step() == 2


def test_frvsr_step_bf16_error_is_bounded():
    """bf16 throughput mode: measured, not parity: HR frames within 2e-2 of the fp32 oracle."""
    F = OT.frvsr_flags(batch_size=2, RNN_N=3, crop_size=16, num_resblock=2)
    S, eng, Rs = run_pair(F, gan=False, act_dtype=torch.bfloat16)
    e = rel_err(eng.gen, frame_major(Rs[-1]["gen_outputs"]))
    assert e < 2e-2, e
    L = eng.losses()
    assert abs(L["l2_content_loss"] - float(dict(zip(Rs[-1]["names"], Rs[-1]["vals"]))["l2_content_loss"])) < 2e-2


def test_bf16_recurrence_with_one_launch_residual_blocks_equals_the_two_launch_recurrence():
    """The latency-regime kernels inside the training step (bf16 mode, configs[1]-shaped crops).
    (a) csrc/resblock_lat.hip alone: the HR frames of all recurrent frames are BIT-identical to the step that runs every residual
        block as two tg_conv_forward launches, and the gradients agree to the noise of their fp32 atomics (bit-identical operands).
    (b) with the HR-tail kernels as well (csrc/hr_fwd_lat.hip, hr_bwd_lat.hip: other accumulation orders than the generic kernels
        they replace): frames and gradients to bf16 rounding."""
    from tecogan_amd.params import damp_values
    F = OT.frvsr_flags(batch_size=2, RNN_N=4, crop_size=32, num_resblock=3)
    x, y = make_batch(F.batch_size, F.RNN_N, F.crop_size)
    res = []
    for blocks, tails, chain in ((True, False, True), (False, False, False), (True, True, True), (True, False, False)):
        eng = TrainEngine(F, DEV, gan=False, act_dtype=torch.bfloat16, seed=7, use_graph=False)
        eng.ps.load(damp_values(eng.ps.state_dict()))
        eng.G.resblock_lat = blocks
        eng.G.resblock_chain = chain          # round 6: the whole trunk of a frame as one persistent launch (csrc/resblock_chain.hip)
        eng.G.hr_fwd_lat = eng.G.hr_bwd_lat = tails
        eng.step(x.to(DEV), y.to(DEV))
        torch.cuda.synchronize()
        assert eng.G._fused_blocks() is blocks and eng.G._chained() is chain
        if chain:
            assert int(eng.G.seq["chain_scratch"][2]) == 0, "a trunk workgroup gave up waiting for a neighbour"
        res.append((eng.gen.clone(), eng.ps.grad.clone(), eng.G.seq["g_in"].clone()))
    (ga, gra, gia), (gb, grb, gib), (gc, grc, gic), (gd, grd, gid) = res
    assert torch.equal(ga, gd), "HR frames differ between the one-launch trunk and the one-launch-per-block recurrence"
    assert torch.equal(gia.view(torch.int16), gid.view(torch.int16)) or rel_err(gia, gid) < 2e-2
    assert torch.equal(ga, gb), "HR frames differ between the one-launch and the two-launch recurrence"
    assert torch.equal(gia.view(torch.int16), gib.view(torch.int16)) or rel_err(gia, gib) < 2e-2   # scatter atomics feed the BPTT
    assert rel_err(gra, grb) < 2e-2
    assert rel_err(gc, gb) < 2e-3, rel_err(gc, gb)                       # HR frames (fp32 outputs of bf16 networks)
    assert rel_err(grc, grb) < 3e-2 and rel_err(gic, gib) < 3e-2, (rel_err(grc, grb), rel_err(gic, gib))


def test_tecogan_step_fp32_parity():
    """Full TecoGAN step (ping-pong, VGG, spatio-temporal D, layer loss, 3 Adams, D-gate) vs the oracle."""
    F = OT.default_flags(batch_size=2, RNN_N=3, crop_size=16, num_resblock=2)
    S, eng, Rs = run_pair(F, gan=True)
    R = Rs[-1]
    check_step(S, eng, R, 1e-3)
    L = eng.losses()
    assert abs(L["All_loss_Gen"] - float(R["gen_loss"])) < 1e-3 * max(1.0, abs(float(R["gen_loss"])))
    assert abs(L["t_balance"] - S.tb) < 1e-5
    assert R["with_D"] is True


def test_tecogan_step_with_deduplicated_vgg_target_pass(monkeypatch):
    """TG_VGGT_DEDUP=1: the VGG features of the ping-pong TARGET frames computed for the RNN_N distinct frames only and
    gathered into sequence order (the mirrored half repeats them) -- same parity against the oracle as the default path, in
    the eager program and in the captured one."""
    monkeypatch.setenv("TG_VGGT_DEDUP", "1")
    F = OT.default_flags(batch_size=2, RNN_N=3, crop_size=16, num_resblock=2)
    for use_graph in (False, True):
        S, eng, Rs = run_pair(F, gan=True, use_graph=use_graph)
        assert eng.vggt_dedup
        check_step(S, eng, Rs[-1], 1e-3)
        L = eng.losses()
        assert abs(L["All_loss_Gen"] - float(Rs[-1]["gen_loss"])) < 1e-3 * max(1.0, abs(float(Rs[-1]["gen_loss"])))


def test_tecogan_three_steps_graph_and_gate():
    """hipGraph replay of the GAN step for 3 steps; the device-side D-gate follows the oracle's decisions."""
    F = OT.default_flags(batch_size=1, RNN_N=4, crop_size=16, num_resblock=1, Dbalance=1e-9)
    S, eng, Rs = run_pair(F, gan=True, steps=3, use_graph=True)
    # tb starts at 0 (< Dbalance) and moves by 1% of t_balance per step; the oracle's decisions:
    gates = [r["with_D"] for r in Rs]
    assert gates[0] is True
    # the engine's D Adam step count (sched[8]) must equal the number of open gates
    assert int(eng.sched[8].item()) == sum(gates), (eng.sched.tolist(), gates)
    # B=1 (tb=2) batch-norm statistics amplify the +-lr Adam noise of earlier steps: loose numeric bound here,
    # the tight multi-step check is test_frvsr_two_steps_graph_replay; this test is about the gate mechanics.
    check_step(S, eng, Rs[-1], 1e-1, adam_slack=2.0)
    assert eng.global_step() == 3


def test_target_lookahead_equals_in_step_target_features():
    """TrainEngine target lookahead: `step(x, y, next_targets=y_next)` puts the NEXT batch's targets through VGG-19 during this
    step's backward phase (segment vggt_next) and the next step starts from the stored features (vggt_pre) -- the same
    features the in-step target pass (vggt) computes, one step earlier.  Three captured steps over three DIFFERENT batches,
    with and without the lookahead: the same frames and losses (the fp32 atomics' noise apart); a validation pass in between
    must not disturb the stored features; a step without an announcement falls back to the in-step pass."""
    F = OT.default_flags(batch_size=1, RNN_N=3, crop_size=16, num_resblock=1)
    batches = [make_batch(1, F.RNN_N, F.crop_size, seed=20 + i) for i in range(4)]
    vx, vy = make_batch(1, F.RNN_N, F.crop_size, seed=99)
    a = TrainEngine(F, DEV, gan=True, act_dtype=torch.float32, seed=42, use_graph=True)
    b = TrainEngine(F, DEV, gan=True, act_dtype=torch.float32, seed=42, use_graph=True)
    assert b.lookahead
    ran = []
    dev = [tuple(t.to(DEV) for t in bt) for bt in batches]              # the announcement is checked by tensor IDENTITY
    for i in range(3):
        x, y = dev[i]
        a.step(x, y)
        announce = dev[i + 1][1] if i != 1 else None                    # step 1 announces nothing: step 2 computes in-step
        if i == 1:
            b.eval_losses(vx.to(DEV), vy.to(DEV))                       # validation between two training steps
        b.step(x, y, next_targets=announce)
        torch.cuda.synchronize()
        ran.append((b._next_ready,))
        assert rel_err(b.gen, a.gen) < 1e-5, (i, rel_err(b.gen, a.gen))
        la, lb = a.losses(), b.losses()
        for k in ("vgg_loss_2", "vgg_loss_5", "l2_content_loss", "t_discrim_loss"):
            assert abs(la[k] - lb[k]) <= 1e-4 * max(1.0, abs(la[k])), (i, k, la[k], lb[k])
    assert ran == [(True,), (False,), (True,)]
    # the promise is checked, not trusted: step 2 announced dev[3][1]; a call that passes ANOTHER tensor (here an equal
    # copy) must compute its target features in-step instead of using the stored ones; the announced object itself may use them
    x, y = dev[3]
    b.step(x, y.clone())
    assert b.used_stored_targets is False
    b.step(x, y, next_targets=dev[0][1])
    b.step(*dev[0])
    assert b.used_stored_targets is True
    torch.cuda.synchronize()
    names = [s["name"] for s in b._segs]
    assert "vggt" in names and "vggt_pre" in names and "vggt_next" in names


def test_tecogan_fading_in_adversarial_weight_stays_captured():
    """lib/Teco.py:379-380: dt_ratio = min(Dt_ratio_max, Dt_ratio_0 + Dt_ratio_add * global_step) scales the adversarial and
    layer losses.  The factor is a device scalar derived from the device-side step counter, so the step is CAPTURED (round 2
    dropped to eager launches whenever Dt_ratio_add != 0) and three replays run with 0.25, 0.5, 0.75."""
    F = OT.default_flags(batch_size=2, RNN_N=3, crop_size=16, num_resblock=2, Dt_ratio_0=0.25, Dt_ratio_add=0.25,
                         Dt_ratio_max=1.0)
    eng = TrainEngine(F, DEV, gan=True, act_dtype=torch.float32, seed=7, use_graph=True)
    assert eng.use_graph
    S = OT.State(F, seed=42, gan=True)
    eng.ps.load(S.P)
    eng.vps.load(S.vgg)
    x, y = make_batch(F.batch_size, F.RNN_N, F.crop_size)
    for k in range(3):
        R = OT.train_step(S, x, y)
        eng.step(x.to(DEV), y.to(DEV))
        torch.cuda.synchronize()
        assert abs(float(eng.dt_ratio.item()) - (0.25 + 0.25 * k)) < 1e-6
        if k == 0:          # first step: exact comparison incl. every gradient (the factor enters G's gradients only)
            check_step(S, eng, R, 1e-3)
    assert eng._segs is not None and all(s["graph"] is not None for s in eng._segs)     # replayed graphs, no eager launches
    mine, ref = eng.losses(), dict(zip(R["names"], [float(v) for v in R["vals"]]))
    for name in ("t_adversarial_loss", "D_layer_loss_sum", "All_loss_Gen"):
        assert abs(mine[name] - ref[name]) <= 0.1 * max(1.0, abs(ref[name])), (name, mine[name], ref[name])


def test_tecogan_no_pingpong_backward_flow_branch():
    """GAN without ping-pong: backward motion comes from an extra FNet call (lib/Teco.py:190-199)."""
    F = OT.default_flags(batch_size=2, RNN_N=3, crop_size=16, num_resblock=1, pingpang=False, vgg_scaling=-0.2)
    S, eng, Rs = run_pair(F, gan=True)
    # 3e-3: with only 2x3 tiny frames a handful of ReLU pre-activations sit within fp32 rounding of 0 and their
    # 0/1 masks flip between summation orders -- a discrete effect on the conv_tran2 weight gradient (1.2e-3).
    check_step(S, eng, Rs[-1], 3e-3)


# ---------------------------------------------------------------------------------------------------------
# round 2: parity at the BASELINE configurations (C2 = configs[1], C3 = configs[2]) with the per-pixel criterion,
# the temporal-only discriminator (Dt_mergeDs=False), and the data-parallel engine itself
# ---------------------------------------------------------------------------------------------------------
from util import assert_close_per_elem, max_rel_err, per_elem_err  # noqa: E402


def check_full_config(F, gan, tag):
    """One fp32 step at a full BASELINE size against the oracle.
    HR frames (the path's OUTPUT): north_star's per-pixel bar, |a-b| <= 1e-3 * max(|b|, 1e-3 max|b|) for EVERY pixel.
    Losses: 1e-3 relative.  Gradients (sums over up to 3e5 pixel products in a different summation order, split-K with
    fp32 atomics, batch-norm backward subtracting sums over 1e5 pixels): relative L2 <= 8e-3 per tensor AND the elements
    within 1e-2 of the tensor's maximum (see the comment at the assertion for the 2 % / 5e-2 allowance), against the oracle
    in FLOAT64.  1e-3 is not attainable in fp32 for every tensor at
    these sizes: the worst tensor, the discriminator's input-conv gradient at C3, is ill-conditioned in fp32 -- the FP32
    ORACLE ITSELF is 5.2e-3 (L2) from its own fp64 run there, and this path measured 2.1e-3 ... 5.4e-3 from the fp64 oracle
    on different runs (the order of the fp32 atomics differs from run to run; since then the batch-norm reductions keep
    double partial sums); all but a handful of the 76 / 132 tensors are below 1e-3 and the worst figures are printed.  (A per-element RELATIVE bound is not meaningful here: an
    element that is ~0 by cancellation of 3e5 terms cannot agree to 1e-6 of the tensor's scale.)
    Weights: seeded xavier, damped (params.damp_values) so that the 10/19-frame recurrence is well conditioned -- with the
    raw xavier init the frame maximum doubles per frame and the fp32 ORACLE itself is 1.6e-2 away from its own fp64 run
    at frame 18 (tests/oracle_conditioning.py), so no fp32 implementation can be held to 1e-3 there."""
    # The oracle runs in float64 here: at these sizes the fp32 oracle's own rounding (batch-norm backward subtracts
    # sums over 1e5 pixels) is of the order of the tolerance, so the fp64 run is the truth both fp32 paths are held to.
    S, eng, Rs = run_pair(F, gan=gan, damp=True, oracle_dtype=torch.float64)
    R = Rs[-1]
    worst = assert_close_per_elem(eng.gen, frame_major(R["gen_outputs"]), 1e-3, 1e-3, what=tag + " gen_outputs")
    L = eng.losses()
    for name, val in zip(R["names"], R["vals"]):
        if name in L:
            assert abs(L[name] - float(val)) <= 1e-3 * max(1e-6, abs(float(val))), (tag, name, L[name], float(val))
    # DERIVED bounds: the same step by the fp32 ORACLE, also measured against the fp64 run -- the rounding a
    # correct fp32 implementation of this graph has on this batch.  Per tensor the HIP path must be within
    # max(floor, FACTOR x the fp32 oracle's own error) of the fp64 truth, in relative L2 AND in max-norm; the blanket caps of
    # round 2 stay as the outer bound for the (at most 2) discriminator tensors whose max-norm is decided by a LeakyReLU mask
    # flip that the fp32 oracle may or may not share (bimodal from run to run, profiles/r02z_c3_repeat.txt).
    from tecogan_amd.params import damp_values
    S0 = OT.State(F, seed=42, gan=gan, dtype=torch.float64)           # the weights run_pair started from (seeded), damped
    S32 = OT.State(F, seed=42, gan=gan, dtype=torch.float32)
    S32.P = type(S0.P)((k, v.float()) for k, v in damp_values(S0.P).items())
    if S0.vgg is not None:
        S32.vgg = type(S0.vgg)((k, v.float()) for k, v in S0.vgg.items())
    x32, y32 = make_batch(F.batch_size, F.RNN_N, F.crop_size)
    P_init = type(S32.P)((k, v.clone()) for k, v in S32.P.items())     # (train_step applies Adam to S32.P in place)
    R32 = OT.train_step(S32, x32, y32)
    check_full_config.last = dict(P_init=P_init, vgg=S32.vgg, x=x32, y=y32, g64={k: v.detach() for k, v in R["grads"].items()},
                                  g32={k: v.detach() for k, v in R32["grads"].items()})
    FACTOR, L2_FLOOR, MX_FLOOR = 1.5, 1e-3, 2e-3
    # The max-norm of a gradient tensor is an extreme-value statistic of ONE summation order (the fp32 atomics add in a
    # run-dependent order): the GAN configuration is therefore stepped by THREE fresh engines from the same weights and the
    # max-norm line is held by the per-tensor MEDIAN of the three (round 4 had widened the allowance from 2 to 4
    # outlier tensors after one run showed three; the bound is back at 2, the statistic is robust instead).  The L2 bound is
    # checked on every run.
    runs = [{n: eng.ps.gview(n).detach().cpu().double() for n in R["grads"]}]
    for _ in range(2 if gan else 0):
        e2 = TrainEngine(F, DEV, gan=gan, act_dtype=torch.float32, seed=7, use_graph=False)
        e2.ps.load(P_init)
        if e2.use_vgg:
            e2.vps.load(S32.vgg)
        e2.step(x32.to(DEV), y32.to(DEV))
        torch.cuda.synchronize()
        runs.append({n: e2.ps.gview(n).detach().cpu().double() for n in R["grads"]})
        del e2
        torch.cuda.empty_cache()
    stats, mask_flips, table = [], [], []
    for name, g in R["grads"].items():
        mine = runs[0][name]
        ref = g.detach().double()
        o32 = R32["grads"][name].detach().double()
        nrm = ref.norm().clamp_min(1e-30)
        l2, l2_o = ((mine - ref).norm() / nrm).item(), ((o32 - ref).norm() / nrm).item()
        for other in runs[1:]:
            l2 = max(l2, ((other[name] - ref).norm() / nrm).item())
        mxs = sorted(max_rel_err(r[name], ref) for r in runs)
        mx, mx_o = mxs[len(mxs) // 2], max_rel_err(o32, ref)
        table.append((l2 / max(L2_FLOOR, FACTOR * l2_o), name, l2, l2_o, mx, mx_o))
        assert l2 <= max(L2_FLOOR, FACTOR * l2_o), "%s gradient %s: relative L2 error %.3g vs the fp64 oracle, fp32 oracle's own %.3g" % (
            tag, name, l2, l2_o)
        if mx > max(MX_FLOOR, FACTOR * mx_o):
            mask_flips.append((name, mx, mx_o))
        pe = per_elem_err(mine, ref, floor=2e-2).max().item()
        bad = ((mine - ref).abs() > 1e-2 * ref.abs().max()).double().mean().item()
        assert l2 < 8e-3 and bad <= 2e-2 and mx < 5e-2, "%s gradient %s: outer caps: L2 %g, max-norm %g (%.2f %% of the elements above 1e-2)" % (
            tag, name, l2, mx, 100 * bad)
        stats.append((l2, pe, mx))
    table.sort(reverse=True)
    print("\n[%s] closest to the derived L2 bound (ratio, tensor, L2 hip, L2 fp32-oracle, max hip, max fp32-oracle):" % tag)
    for row in table[:4]:
        print("    %.2f %s %.2e %.2e %.2e %.2e" % row)
    # The MAX-norm compares one summation order with another (fp32 atomics in a run-dependent order here, a fixed but different
    # order in the fp32 oracle): by the median of three runs at most TWO tensors may leave the 1.5x line (round 3's bound), only
    # discriminator tensors (LeakyReLU / batch-norm backward behind 1e5-pixel sums: a mask flip the fp32 oracle may or may not
    # share), and never beyond 3x what the fp32 oracle itself shows on that tensor.  Single runs showed 0-2 such tensors in rounds
    # 2-3 and three in one run of round 4 (1.8x / 1.7x / 1.55x; profiles/r04z_pytest_gpu.log).
    print("[%s] max-norm beyond the derived line (median of %d runs): %s" % (tag, len(runs), [(n.split("/")[-3:], "%.2e" % a, "%.2e" % b) for n, a, b in mask_flips]))
    # Round 5: the COUNT of tensors above the 1.5x line is itself a coin flip -- every discriminator tensor's ratio (this path's
    # max-norm error over the fp32 oracle's, both against fp64) scatters around ~1.2 because both are one draw of rounding noise
    # amplified by the batch-norm backward pass (L2 of the ORACLE 5e-3 there), so with 14 such tensors "at most two above 1.5"
    # failed in 2 of the 4 full-suite sessions of rounds 4-5 with three tensors at 1.55-1.8x and NO code change in the fp32 path
    # (profiles/r04z_, r04zz_, r05a_, r05g_pytest_gpu.log).  What a systematic loss of accuracy would do is move the whole
    # distribution, so that is what is held now: the MEDIAN ratio over the discriminator's tensors within 1.35 (measured 1.02-1.18 in
    # sessions H, Z, ZZ, profiles/r05*_pytest*.log: the same code flagged three tensors in sessions G, Z and two in H, ZZ), at most two tensors
    # beyond 2x, none beyond 3x, and every non-discriminator tensor on the 1.5x line as before (the L2 bound above, 1.5x per tensor
    # on every run, is unchanged and is the tight one).
    d_ratios = sorted((row[4] / max(row[5], MX_FLOOR / FACTOR), row[1].split("/")[-3] + "/" + row[1].split("/")[-1])
                      for row in table if row[1].startswith("tdiscriminator"))
    print("[%s] discriminator max-norm ratios (this path / fp32 oracle): %s" % (tag, ["%.2f %s" % r for r in d_ratios]))
    assert all(n.startswith("tdiscriminator") and mx <= max(MX_FLOOR, 3.0 * mx_o) for n, mx, mx_o in mask_flips), \
        "%s: max-norm beyond the derived bound on %s" % (tag, mask_flips)
    if d_ratios:
        med = d_ratios[len(d_ratios) // 2][0]
        assert med <= 1.35 and sum(r > 2.0 for r, _ in d_ratios) <= 2, \
            "%s: discriminator max-norm ratios: median %.2f, %s" % (tag, med, d_ratios[-4:])
    print("\n[%s] gen per-pixel err %.2e; %d gradient tensors vs the fp64 oracle: worst L2 %.2e (%d above 1e-3), worst max-norm "
          "%.2e, worst per-element (floor 2e-2) %.2e" %
          (tag, worst, len(stats), max(s[0] for s in stats), sum(s[0] > 1e-3 for s in stats), max(s[2] for s in stats),
           max(s[1] for s in stats)))
    return S, eng, R


def test_frvsr_step_fp32_parity_at_baseline_config_C2():
    """BASELINE.json configs[1]: runGan.py 4, B=4, RNN_N=10, 32x32 LR, num_resblock=10."""
    check_full_config(OT.frvsr_flags(), gan=False, tag="C2")


def test_tecogan_step_fp32_parity_at_baseline_config_C3():
    """BASELINE.json configs[2]: runGan.py 3, B=4, RNN_N=10 (19 frames with ping-pong), num_resblock=16, Dst + VGG."""
    S, eng, R = check_full_config(OT.default_flags(), gan=True, tag="C3")
    assert R["with_D"] is True and int(eng.sched[8].item()) == 1
    del eng
    torch.cuda.empty_cache()
    _c3_frozen_max_norm(check_full_config.last)


C3_FROZEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c3_det_max_norm.json")


def _c3_frozen_max_norm(last):
    """the max-norm criterion WITHOUT run-to-run noise.  The same C3 step in the ordered-reduction parity mode
    (TG_DETERMINISTIC=1, a subprocess: the switch is read once per process) is bit-reproducible, so every gradient tensor's
    max-norm error against the fp64 oracle is a fixed number: recorded once in tests/golden/c3_det_max_norm.json (tools/c3_repeat.py
    --ratios) together with the fp32 oracle's own, and held here -- (1) reproduced to 2 % (the fp64 oracle's CPU summation order is
    the only free variable), (2) on the frozen numbers: every tensor within max(2e-3, 3 x the fp32 oracle's error), the
    discriminator's median ratio within 1.35, at most two of its tensors beyond 2 x.  A systematic loss of accuracy moves these
    numbers; summation-order noise cannot."""
    import json
    import subprocess
    import sys
    import tempfile
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "c3_repeat.py")
    with tempfile.TemporaryDirectory() as tmp:
        blob, out = os.path.join(tmp, "c3.pt"), os.path.join(tmp, "ratios.json")
        torch.save(last, blob)
        env = dict(os.environ, TG_DETERMINISTIC="1")
        r = subprocess.run([sys.executable, tool, "--ratios", blob, out], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        got = json.load(open(out))
    if not os.path.exists(C3_FROZEN):
        dump = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "c3_det_max_norm.json")
        os.makedirs(os.path.dirname(dump), exist_ok=True)
        json.dump(got, open(dump, "w"), indent=0, sort_keys=True)
        pytest.skip("tests/golden/c3_det_max_norm.json absent: recorded to gpurun_out/c3_det_max_norm.json")
    want = json.load(open(C3_FROZEN))
    assert set(got) == set(want)
    for name, (l2, mx, l2_o, mx_o) in want.items():
        g = got[name]
        assert abs(g[1] - mx) <= 0.02 * mx + 1e-7 and abs(g[0] - l2) <= 0.02 * l2 + 1e-7, (name, g, want[name])
        assert mx <= max(2e-3, 3.0 * mx_o) and l2 <= max(1e-3, 1.5 * l2_o), (name, want[name])
    dr = sorted(v[1] / max(v[3], 2e-3 / 1.5) for k, v in want.items() if k.startswith("tdiscriminator"))
    print("[C3, deterministic mode] discriminator max-norm ratios (frozen): median %.2f, max %.2f" % (dr[len(dr) // 2], dr[-1]))
    assert dr[len(dr) // 2] <= 1.35 and sum(r > 2.0 for r in dr) <= 2, dr


def test_bf16_mode_error_at_baseline_config_C2():
    """The timed bf16 mode at the full C2 size: measured, stated bound (not parity): HR frames within 3e-2 of the oracle
    relative to the frame maximum, content loss within 2 %."""
    F = OT.frvsr_flags()
    S, eng, Rs = run_pair(F, gan=False, act_dtype=torch.bfloat16, damp=True)
    e = max_rel_err(eng.gen, frame_major(Rs[-1]["gen_outputs"]))
    ref = float(dict(zip(Rs[-1]["names"], Rs[-1]["vals"]))["l2_content_loss"])
    print("\n[C2 bf16] gen max-rel err %.3e, content loss %.5f vs %.5f" % (e, eng.losses()["l2_content_loss"], ref))
    assert e < 3e-2, e
    assert abs(eng.losses()["l2_content_loss"] - ref) < 2e-2 * ref


def test_bf16_mode_error_at_baseline_config_C3():
    """The TIMED mode (bf16 activations / fp32 master weights) at the full configs[2] size against the fp32 mode of the same
    engine (held to the fp64 oracle by test_tecogan_step_fp32_parity_at_baseline_config_C3): one step from identical damped
    weights and batch.  Stated, measured bounds -- not parity (profiles/r04b_bf16_error_table.txt, tools/bf16_error_table.py):
      * HR frames: relative L2 <= 6e-3 (measured 3.3e-3), every loss scalar within 1 %;
      * gradients per optimiser scope, relative L2: generator <= 2e-2 (9.6e-3), FNet <= 1e-1 (6.6e-2), D <= 1.6e-1 (1.06e-1);
      * the gradient of D's LAST layer (no activation decision between it and the loss) <= 1e-2 (2.6e-3): bf16 accumulation is
        not what separates the modes.  What does: every ReLU / LeakyReLU / max-pool decision whose pre-activation lies within
        the forward difference of the two modes (1e-2 after 5-35 bf16 layers) flips, and a flipped unit contributes its FULL
        gradient as error -- a fraction f of flipped units costs ~sqrt(f) in relative L2 (VGG's input gradient: 2.6e-1 after
        16 ReLU layers and 4 pools).  Both gradients are exact gradients of networks 1e-2 apart; no storage format short of
        fp32 activations removes this, and bench.py's trajectory control shows the modes' loss curves differ by what an fp32
        run differs from itself under a one-time perturbation of bf16's size."""
    from tecogan_amd.params import damp_values
    F = OT.default_flags()
    x, y = make_batch(F.batch_size, F.RNN_N, F.crop_size)
    res = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        eng = TrainEngine(F, DEV, gan=True, act_dtype=dt, seed=7, use_graph=False)
        eng.ps.load(damp_values(eng.ps.state_dict()))
        eng.step(x.to(DEV), y.to(DEV))
        torch.cuda.synchronize()
        grads = {sc: eng.ps.scope_slice(sc, eng.ps.grad).double().cpu() for sc in eng.ps.scope_range}
        last = eng.ps.gview("tdiscriminator/discriminator_unit/dense_layer_2/dense/kernel").double().cpu().clone()
        res[name] = (eng.gen.double().cpu(), eng.losses(), grads, last)
        del eng
        torch.cuda.empty_cache()
    (gf, lf, grf, lastf), (gb, lb, grb, lastb) = res["f32"], res["bf16"]
    rl2 = lambda a, b: float((a - b).norm() / b.norm())                           # noqa: E731
    e_gen = rl2(gb, gf)
    e_scope = {sc: rl2(grb[sc], grf[sc]) for sc in grf}
    e_last = rl2(lastb, lastf)
    print("\n[C3 bf16 vs fp32] frames %.2e, gradients %s, D last layer %.2e" % (e_gen, {k: "%.2e" % v for k, v in e_scope.items()}, e_last))
    assert e_gen < 6e-3, e_gen
    for k, v in lf.items():
        if k in lb and abs(v) > 1e-6 and k not in ("t_balance", "t_balance_now"):
            assert abs(lb[k] - v) <= 1e-2 * abs(v), (k, lb[k], v)
    assert e_scope["generator"] < 2e-2 and e_scope["fnet"] < 1e-1 and e_scope["tdiscriminator"] < 1.6e-1, e_scope
    assert e_last < 1e-2, e_last


def test_bf16_latency_kernels_on_vs_off_at_baseline_config_C3():
    """the TIMED code path under a tight test at the BASELINE size.  The fp32 parity tests above cannot run the
    latency-regime kernels (csrc/resblock_lat.hip, hr_fwd_lat.hip, hr_bwd_lat.hip are bf16-only), so at configs[2] (B=4, 19
    frames, num_resblock=16, D + VGG + ping-pong) the bf16 engine runs one step three times from identical damped weights:
      A  all latency kernels ON (the timed configuration);
      B  residual blocks ON, HR tails on the generic kernels;
      C  everything on the generic tg_conv_forward launches -- the code the fp32 C3 parity test exercises (in fp32).
    B vs C: the one-launch block is built to be BIT-identical to its two launches -> all 19 HR frames bit-equal, the recurrent
    input-gradient buffer bit-equal or within scatter-atomics noise, gradients within fp32-atomics noise.
    A vs C: the HR-tail kernels accumulate in another order -> frames to bf16 rounding (2e-3 of the frame range, relative L2
    1e-3), every loss scalar within 0.5 %, gradients per optimiser scope within the decision-flip bounds stated below (a
    1e-3 forward difference flips ReLU / LeakyReLU / max-pool decisions in VGG-19 and D: the mechanism of
    test_bf16_mode_error_at_baseline_config_C3 at a tenth of its forward difference)."""
    from tecogan_amd.params import damp_values
    F = OT.default_flags()
    x, y = make_batch(F.batch_size, F.RNN_N, F.crop_size)
    res = {}
    for tag, blocks, tails in (("A", True, True), ("B", True, False), ("C", False, False), ("C2", False, False)):
        eng = TrainEngine(F, DEV, gan=True, act_dtype=torch.bfloat16, seed=7, use_graph=False)
        eng.ps.load(damp_values(eng.ps.state_dict()))
        eng.G.resblock_lat = blocks
        eng.G.hr_fwd_lat = eng.G.hr_bwd_lat = tails
        eng.step(x.to(DEV), y.to(DEV))
        torch.cuda.synchronize()
        assert eng.G._fused_blocks() is blocks
        res[tag] = (eng.gen.clone(), eng.losses(), {sc: eng.ps.scope_slice(sc, eng.ps.grad).double().cpu() for sc in eng.ps.scope_range},
                    eng.G.seq["g_in"].clone())
        del eng
        torch.cuda.empty_cache()
    rl2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))     # noqa: E731
    (ga, la, gra, gia), (gb, lb, grb, gib), (gc, lc, grc, gic) = res["A"], res["B"], res["C"]
    gc2, _, grc2, gic2 = res["C2"]
    assert torch.equal(gb, gc) and torch.equal(gc2, gc), "C3: HR frames differ between the one-launch and the two-launch residual blocks"
    # run-to-run NOISE of the bf16 step itself (C vs C2: the same code twice): batch-norm and loss sums accumulate with fp32
    # atomics in a run-dependent order, the sums are rounded to bf16 activations, decisions flip -- 1e-2 on D's gradient and on the
    # seeds of the BPTT (profiles/r05b_pytest_quick.log).  B vs C is held to that noise, not to a constant.
    n_g = {sc: rl2(grc2[sc], grc[sc]) for sc in grc}
    n_gi = (rel_err(gic2, gic), rl2(gic2, gic))
    e_bc = {sc: rl2(grb[sc], grc[sc]) for sc in grc}
    e_ac = {sc: rl2(gra[sc], grc[sc]) for sc in grc}
    print("\n[C3 bf16 latency kernels] run-to-run noise: gradients %s, g_in max %.1e / L2 %.1e; blocks only vs generic: frames bit-equal, "
          "gradients %s, g_in max %.1e / L2 %.1e; all ON vs generic: frames max %.2e / L2 %.2e, gradients %s"
          % ({k: "%.1e" % v for k, v in n_g.items()}, n_gi[0], n_gi[1], {k: "%.1e" % v for k, v in e_bc.items()}, rel_err(gib, gic),
             rl2(gib, gic), rel_err(ga, gc), rl2(ga, gc), {k: "%.1e" % v for k, v in e_ac.items()}))
    assert torch.equal(gib.view(torch.int16), gic.view(torch.int16)) or \
        (rel_err(gib, gic) <= 3.0 * n_gi[0] + 1e-3 and rl2(gib, gic) <= 3.0 * n_gi[1] + 1e-4), (rel_err(gib, gic), rl2(gib, gic), n_gi)
    assert all(e_bc[sc] <= 3.0 * n_g[sc] + 1e-4 for sc in e_bc), (e_bc, n_g)
    assert rel_err(ga, gc) < 2e-3 and rl2(ga, gc) < 1e-3, (rel_err(ga, gc), rl2(ga, gc))
    for k, v in lc.items():
        if k in la and abs(v) > 1e-6 and k not in ("t_balance", "t_balance_now"):
            assert abs(la[k] - v) <= 5e-3 * abs(v), (k, la[k], v)
    assert e_ac["generator"] < 2e-2 and e_ac["fnet"] < 6e-2 and e_ac["tdiscriminator"] < 1e-1, e_ac


def test_tecogan_temporal_only_discriminator_Dt_mergeDs_false():
    """lib/Teco.py:246-250,269-272,423-424: D sees only the 9 warped channels, centre-cropped to (4h - 2 off)^2, and its
    learning rate is 0.3 x.  (The reference's own branch cannot run -- discriminator_F returns a tuple there and the layer
    loss reads undefined names -- so the oracle states the evident intent: same D, 9-channel input, layer loss on.)"""
    F = OT.default_flags(batch_size=2, RNN_N=3, crop_size=16, num_resblock=1, Dt_mergeDs=False)
    S, eng, Rs = run_pair(F, gan=True)
    assert eng.ps.entries["tdiscriminator/discriminator_unit/input_stage/conv/Conv/weights"]["shape"] == (3, 3, 9, 64)
    check_step(S, eng, Rs[-1], 3e-3)
    assert abs(float(eng.hyper[0, 5].item()) - 0.3 * F.learning_rate) < 1e-9     # D's base learning rate


def _dp_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    F = OT.frvsr_flags(batch_size=1, RNN_N=3, crop_size=16, num_resblock=2)
    eng = TrainEngine(F, "cuda:0", gan=False, act_dtype=torch.float32, seed=42, process_group=dist.group.WORLD, use_graph=True)
    x, y = make_batch(2, F.RNN_N, F.crop_size, seed=77)
    for _ in range(2):
        eng.step(x[rank:rank + 1].cuda(), y[rank:rank + 1].cuda())
    torch.cuda.synchronize()
    # numpy arrays travel through the queue by value (torch tensors would be handed over as file descriptors of a
    # resource-sharer socket that disappears when this process exits)
    q.put((rank, eng.exchange_mode, {k: v.numpy().copy() for k, v in eng.ps.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_engine_world2_equals_single_process_double_batch():
    """TrainEngine(process_group=...) with two ranks (sharing this GPU, gloo): per-rank batch 1 each, gradient all-reduce
    with 1/world folded into Adam, two-graph split.  After two steps both ranks hold the weights of ONE process
    stepping on the concatenated batch of 2 (every FRVSR loss is a batch mean, lib/Teco.py:322,331)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == "eager-split"
    F = OT.frvsr_flags(batch_size=2, RNN_N=3, crop_size=16, num_resblock=2)
    ref = TrainEngine(F, DEV, gan=False, act_dtype=torch.float32, seed=42, use_graph=True)
    x, y = make_batch(2, F.RNN_N, F.crop_size, seed=77)
    for _ in range(2):
        ref.step(x.to(DEV), y.to(DEV))
    torch.cuda.synchronize()
    lr = F.learning_rate
    for name, w in ref.ps.state_dict().items():
        for rank in (0, 1):
            d = (torch.from_numpy(res[rank][2][name]) - w).abs().max().item()
            # Adam's +-lr sign noise on ~0 gradients (see check_step): 2 steps x 2 lr, plus 1e-3 of the tensor's scale
            assert d <= 1e-3 * w.abs().max().item() + 4.0 * lr, (name, rank, d)
        assert (res[0][2][name] == res[1][2][name]).all(), "ranks diverged on " + name


def test_captured_rccl_exchange_single_rank_group():
    """The captured-exchange path (RCCL all-reduce nodes inside the step's hipGraph, issued on the communication stream)
    on a one-rank RCCL group: same result as the engine without a process group."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(31000 + os.getpid() % 2000))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        F = OT.default_flags(batch_size=1, RNN_N=3, crop_size=16, num_resblock=1)
        x, y = make_batch(1, F.RNN_N, F.crop_size, seed=5)
        a = TrainEngine(F, DEV, gan=True, act_dtype=torch.float32, seed=42, use_graph=True)
        b = TrainEngine(F, DEV, gan=True, act_dtype=torch.float32, seed=42, use_graph=True, process_group=dist.group.WORLD)
        b.world, b.exchange_mode, b.segmented = 1, "captured", True    # force the collective segments although world == 1
        b.comm_stream = b.streams["C"] = torch.cuda.Stream()
        for _ in range(2):
            a.step(x.to(DEV), y.to(DEV))
            b.step(x.to(DEV), y.to(DEV))
        torch.cuda.synchronize()
        assert b.exchange_mode == "captured", "capture of the RCCL collectives fell back to the eager split"
        # the node census of the captured exchange segments (tg_graph_node_count): a ONE-rank communicator's all-reduce is elided
        # by RCCL, at most the 1/world scaling kernel of the balance scalar remains -- which is why an engine with real ranks
        # (require_exchange_nodes) refuses such a capture instead of replaying nothing
        assert set(b.exchange_nodes) == {"ar_d", "ar_g", "ar_f"} and not b.require_exchange_nodes, b.exchange_nodes
        assert b.exchange_nodes["ar_g"][1] <= 0 and b.exchange_nodes["ar_f"][1] <= 0, b.exchange_nodes
        b2 = TrainEngine(F, DEV, gan=True, act_dtype=torch.float32, seed=42, use_graph=True, process_group=dist.group.WORLD)
        b2.world, b2.exchange_mode, b2.segmented, b2.require_exchange_nodes = 1, "captured", True, True
        b2.comm_stream = b2.streams["C"] = torch.cuda.Stream()
        with pytest.raises(RuntimeError, match="hold no kernel node"):
            b2.step(x.to(DEV), y.to(DEV))
        torch.cuda.synchronize()
        for name, w in a.ps.state_dict().items():
            d = (b.ps.view(name).cpu() - w).abs().max().item()
            assert d <= 1e-3 * w.abs().max().item() + 4.0 * F.learning_rate, (name, d)
    finally:
        dist.destroy_process_group()


def test_captured_exchange_segments_carry_nodes_standin_world2():
    """SURVEY 8e on ONE GPU: the 2-rank program (communication stream, captured exchange segments ar_d / ar_g / ar_f,
    t_balance reduced before the D gate, 1/world folded into Adam) with every all-reduce replaced by the in-place `x *= 2`
    kernel a sum over two identical ranks amounts to (TrainEngine(standin_world=2)).  Unlike a one-rank RCCL communicator --
    RCCL elides the kernel there, so the captured graph was EMPTY -- every communication segment here holds real nodes that
    are replayed on the communication stream, and the result equals the one-rank engine's."""
    import warnings
    F = OT.default_flags(batch_size=1, RNN_N=3, crop_size=16, num_resblock=1)
    x, y = make_batch(1, F.RNN_N, F.crop_size, seed=5)
    a = TrainEngine(F, DEV, gan=True, act_dtype=torch.float32, seed=42, use_graph=True)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        b = TrainEngine(F, DEV, gan=True, act_dtype=torch.float32, seed=42, use_graph=True, standin_world=2)
        for _ in range(3):
            a.step(x.to(DEV), y.to(DEV))
            b.step(x.to(DEV), y.to(DEV))
        torch.cuda.synchronize()
    assert not [w for w in rec if "empty" in str(w.message).lower()], [str(w.message) for w in rec]
    assert b.exchange_mode == "captured" and b.world == 2
    assert b.exchange_segments == ["ar_d", "ar_g", "ar_f"]
    assert all(kern >= 1 for _, kern in b.exchange_nodes.values()) and b.exchange_nodes["ar_d"][1] >= 3, b.exchange_nodes
    csegs = [s for s in b._segs if s["skey"] == "C"]
    assert [s["name"] for s in csegs] == ["ar_d", "ar_g", "ar_f"] and all(s["graph"] is not None for s in csegs)
    upd = [s for s in b._segs if s["name"] == "update"][0]
    assert {"ar_d", "ar_g", "ar_f"} <= set(upd["deps"])
    assert b.allreduce_bytes() == 4 * (b.ps.flat.numel()) + 4
    assert int(a.sched[8].item()) == int(b.sched[8].item())          # same number of (gated) D updates
    for name, w in a.ps.state_dict().items():
        d = (b.ps.view(name).cpu() - w).abs().max().item()
        assert d <= 1e-3 * w.abs().max().item() + 6.0 * F.learning_rate, (name, d)


def test_host_jitter_before_exchange_segments_standin_world8():
    """VERDICT r4 item 7 / Weak 7: the exchange segments are launched JUST IN TIME -- the host waits for their dependencies and
    only then enqueues them -- so on an 8-GPU node every collective sits behind a host-side wait of its rank, and per-rank host
    jitter lands in front of every all-reduce.  The untested part of the design, made to bite on one GPU: the 8-rank program
    (TrainEngine(standin_world=8): communication stream, captured ar_d / ar_g / ar_f segments, 1/8 folded into Adam) at the
    TIMED size (configs[2], bf16) with a random 0-200 us host sleep injected before every communication-segment launch
    (SegmentRunner.launch_jitter).  ar_d is issued after D's own-gradient passes and has the whole BPTT (2.3 ms) to hide
    under, ar_g / ar_f FNet's backward pass: the step time must stay within 3 % (+20 us), the result unchanged."""
    import random
    import time
    from tecogan_amd.params import damp_values
    F = OT.default_flags()
    x, y = (t.to(DEV) for t in make_batch(F.batch_size, F.RNN_N, F.crop_size))

    def build():
        e = TrainEngine(F, DEV, gan=True, act_dtype=torch.bfloat16, seed=7, use_graph=True, standin_world=8)
        e.ps.load(damp_values(e.ps.state_dict()))
        e.set_batch(x, y)
        return e

    def timed(e, steps=20, blocks=4):
        best = 1e9
        for _ in range(blocks):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                e.step(next_targets=True)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps * 1e3)
        return best

    a, b = build(), build()
    for e in (a, b):
        for _ in range(3):
            e.step(next_targets=True)
    torch.cuda.synchronize()
    assert b.exchange_segments == ["ar_d", "ar_g", "ar_f"] and b.world == 8
    # the RESULT first: one step with the maximal sleep before every exchange segment against one plain step, both engines in
    # the same state (three steps in) -- equal to the noise of the fp32 atomics
    b.launch_jitter = lambda name: 200e-6 if name.startswith("ar_") else 0.0
    a.step(next_targets=True)
    b.step(next_targets=True)
    b.launch_jitter = None
    torch.cuda.synchronize()
    assert rel_err(b.gen, a.gen) < 5e-3, rel_err(b.gen, a.gen)
    la, lb = a.losses(), b.losses()
    for k in ("l2_content_loss", "l2_warp_loss", "t_discrim_loss", "vgg_loss_3"):
        assert abs(la[k] - lb[k]) <= 2e-3 * max(abs(la[k]), 1e-6), (k, la[k], lb[k])
    assert rel_err(b.ps.grad, a.ps.grad) < 5e-2
    t_plain = min(timed(a), timed(b))
    rng, slept = random.Random(1), []

    def jitter(name):
        assert name.startswith("ar_") or name in ("vggt", "vggt_pre", "dreal", "vgg_0", "vgg_1", "vggt_next", "wgrad", "down"), name
        if not name.startswith("ar_"):
            return 0.0
        slept.append(rng.uniform(0.0, 200e-6))
        return slept[-1]
    b.launch_jitter = jitter
    t_jit = timed(b)
    b.launch_jitter = None
    t_plain = min(t_plain, timed(b), timed(a))           # (the plain steps again AFTER the jittered ones: a box that drifts by tenths of
    b.launch_jitter = jitter                             #  a millisecond between the two measurements is not the launcher's doing)
    t_jit = min(t_jit, timed(b))
    b.launch_jitter = None
    per_step = sum(slept) / max(len(slept), 1) * 3
    print("\n[stand-in world 8, configs[2] bf16] step %.3f ms plain, %.3f ms with 0-200 us host jitter before each of the 3 exchange "
          "segments (%.0f us of sleep per step)" % (t_plain, t_jit, per_step * 1e6))
    assert len(slept) >= 2 * 3 * 80
    # What is held: injected host sleep is never AMPLIFIED -- the step grows by at most the sleep itself (+ 30 us of timing noise).
    # Most of it cannot hide by construction: `ar_g` and `ar_f` are launched when the generator's / FNet's gradients exist -- the tail
    # of the step -- and `update` needs both at once, so two of the three sleeps (~200 of 296 us) sit on the critical path whatever
    # the launcher does; `ar_d`'s sleep (it overlaps the whole BPTT) hides since the exchange segments are launched from their own
    # thread (segments.SegmentRunner.comm_thread, round 5: 296 us of sleep cost 284 us before, 232 us after; the round-4 form of this
    # line, "within 3 %", held only while the step was slower than 8.2 ms).  Real launch jitter is tens of microseconds per segment.
    print("    hidden fraction of the injected sleep: %.2f" % (1.0 - (t_jit - t_plain) / (per_step * 1e3)))
    assert t_jit - t_plain <= per_step * 1e3 + max(0.03, 0.01 * t_plain), (t_plain, t_jit, per_step)     # (+ 1 % of the step: box noise)
    # ... and the part that CAN hide does (the line above passes even if none of the sleep is hidden): a fixed 200 us
    # sleep in front of `ar_d` alone -- it is launched right after D's own-gradient passes and `update`, its only consumer, is a
    # whole BPTT (2 ms) away -- must cost less than half of itself
    b.launch_jitter = lambda name: 200e-6 if name == "ar_d" else 0.0
    t_d = timed(b)
    b.launch_jitter = None
    t_plain = min(t_plain, timed(b))                     # (plain steps on both sides of the measurement: box drift is not the launcher's)
    b.launch_jitter = lambda name: 200e-6 if name == "ar_d" else 0.0
    t_d = min(t_d, timed(b))
    b.launch_jitter = None
    hidden_d = 1.0 - (t_d - t_plain) / 0.2
    print("    200 us before ar_d alone: step %.3f ms, hidden fraction %.2f" % (t_d, hidden_d))
    assert hidden_d >= 0.5, (t_plain, t_d)
    assert all(v == v for v in b.losses().values())


def test_validation_pass_issues_no_collective_and_leaves_training_state_untouched():
    """ADVICE r3 (main.py validation on rank 0, reference main.py:391-402): `eval_losses` runs the step's program without the
    update segment -- and, in the captured multi-rank program, without ANY exchange segment: an all-reduce issued by one rank
    alone would pair with the other ranks' next training step and shift every later collective by one.  2-rank stand-in
    program on one GPU; the weights, Adam state and schedule are unchanged by the validation pass and the next training
    step equals the one of an engine that never validated."""
    F = OT.default_flags(batch_size=1, RNN_N=3, crop_size=16, num_resblock=1)
    x, y = make_batch(1, F.RNN_N, F.crop_size, seed=5)
    vx, vy = make_batch(1, F.RNN_N, F.crop_size, seed=6)
    engs = [TrainEngine(F, DEV, gan=True, act_dtype=torch.float32, seed=42, use_graph=True, standin_world=2) for _ in range(2)]
    for e in engs:
        e.step(x.to(DEV), y.to(DEV))
    torch.cuda.synchronize()
    a, b = engs
    before = [t.clone() for t in (b.ps.flat, b.ps.m, b.ps.v, b.sched, b.hyper)]
    issued = []
    orig = b._sum_all_reduce
    b._sum_all_reduce = lambda t: (issued.append(t.numel()), orig(t))[1]
    V = b.eval_losses(vx.to(DEV), vy.to(DEV))
    b._sum_all_reduce = orig
    assert issued == [] and b.exchange_segments == [], "the validation pass issued %d collectives" % len(issued)
    assert all(v == v for v in V.values()) and "l2_content_loss" in V
    for t, c in zip((b.ps.flat, b.ps.m, b.ps.v, b.sched, b.hyper), before):
        assert torch.equal(t, c)
    for e in engs:
        e.step(x.to(DEV), y.to(DEV))
    torch.cuda.synchronize()
    # (gradients and frames, not weights: two engines differ by the noise of their fp32 atomics, and Adam's first updates are
    #  lr * sign(g) -- a gradient element near zero moves its weight by a whole step either way)
    eg, ef = rel_err(b.ps.grad, a.ps.grad), rel_err(b.gen, a.gen)
    assert eg < 1e-3 and ef < 1e-4, (eg, ef)
    assert int(a.sched[0].item()) == int(b.sched[0].item()) == 2


def test_bench_gpus_2_on_one_device_runs_two_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself; TG_DIST_BACKEND=gloo lets
    them share this box's single GPU (plumbing: eager-split exchange over gloo).  The line must say n_gpus 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["TG_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "frvsr", "--steps", "3", "--warmup", "1",
                        "--no-sub", "--no-roofline", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and line["config"]["global_batch"] == 8
    assert line["config"]["exchange"].startswith("eager-split") and line["config"]["allreduce_bytes_per_step"] > 0
    # the N > 1 line explains its exchange: per-segment timing from device stamps
    tl = line["config"]["exchange_timeline"]
    assert "error" not in tl, tl
    assert set(tl) >= {"step_ms", "segments", "compute"} and "exchange" in tl["segments"], tl
    assert {"start_ms", "ms", "under_ms", "exposed_ms"} <= set(tl["segments"]["exchange"])


def test_deterministic_parity_mode_is_bit_reproducible():
    """TG_DETERMINISTIC=1 (csrc/common.h): every accumulation with floating-point atomics runs in a fixed order -- reductions as
    one workgroup, weight gradients without split-K, the scatter kernels as one wavefront -- so that three fresh engines produce
    bit-identical frames, loss slots, gradients and post-Adam weights (the switch is read once per process: subprocess), while
    the default mode is allowed to differ.  A regression is then distinguishable from summation-order noise."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "c3_repeat.py")
    env = dict(os.environ, TG_DETERMINISTIC="1")
    for cfg, runs in (("small", "3"), ("c3", "2")):            # c3 = BASELINE configs[2]: B=4 x 19 frames, D + VGG
        r = subprocess.run([sys.executable, tool, "--config", cfg, "--runs", runs], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        assert r.stdout.strip().splitlines()[-1] == "IDENTICAL", (cfg, r.stdout[-1500:])
