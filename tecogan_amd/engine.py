"""Training engine: the FRVSR / TecoGAN step of reference lib/Teco.py:77-517 as ONE explicit kernel
program (forward, hand-written backward, RCCL gradient all-reduce, device-side schedule, fused
TF-Adam), captured once into a hipGraph and replayed per step -- no tracing compiler, no autograd tape,
no host round trip inside a step (the tf.cond D-gate is a device-side predicate).

Data layout: sequences are frame-major `[T, B, ...]` NHWC so that each recurrent frame, each FNet pair
and each flow slab is a contiguous `[B, ...]` block.  Semantics that differ from the reference on
purpose: all gradients are taken from the pre-update weights, then D (gate permitting), G and FNet
are applied (the TF1 graph has an ordering race there).
"""
import os
from collections import OrderedDict

import torch

from . import kernels as K
from . import promise
from .nets import FNET_CPAD, GEN_CPAD, VGG_CPAD, VGG_TAPS, Discriminator, FNet, Generator, VGG19
from .parallel import ExchangeMixin
from .params import (DIS_BLOCKS, ParamStore, discriminator_spec, fnet_spec, generator_spec, init_values, pad8,
                     vgg_spec)
from .streams import shared_stream as _shared_stream
from .segments import SegmentRunner, plan_launch_order  # noqa: F401  (plan_launch_order: re-exported for tools/)

LOSS_NAMES = ["l2_content_loss", "l2_warp_loss", "PingPang", "vgg_loss_2", "vgg_loss_3", "vgg_loss_4", "vgg_loss_5",
              "t_adversarial_loss", "t_discrim_loss", "t_balance", "t_discrim_real_output", "t_discrim_fake_output",
              "D_layer_0_loss", "D_layer_1_loss", "D_layer_2_loss", "D_layer_3_loss"]
LI = {n: i for i, n in enumerate(LOSS_NAMES)}




class TrainEngine(SegmentRunner, ExchangeMixin):
    def __init__(self, flags, device="cuda", gan=True, act_dtype=torch.float32, seed=42, process_group=None,
                 use_graph=True, standin_world=0):
        """standin_world=W (tests, one GPU, no process group): run the W-rank program -- communication stream, captured
        exchange segments, 1/W folded into Adam -- with every all-reduce replaced by an in-place `x *= W` kernel on the
        communication stream (what a sum over W ranks holding identical gradients gives).  The result must equal the
        one-rank engine's, and every exchange segment carries real graph nodes (RCCL itself elides the kernel for a
        one-rank communicator, so a one-rank group cannot test that)."""
        F = self.F = flags
        self.dev, self.gan, self.act_dtype = torch.device(device), gan, act_dtype
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.standin = int(standin_world) if process_group is None else 0
        if self.standin > 1:
            self.world = self.standin
        # Gradient exchange.  "captured": the RCCL all-reduces are nodes of the step's hipGraph, issued on a
        # side stream as soon as a scope's gradients are final (D right after its backward passes, G after the
        # sequence-wide weight gradients) so they overlap the rest of the backward pass; "eager-split": compute graph,
        # eager collectives, update graph (any backend that cannot be captured, e.g. gloo in the plumbing tests).
        self.exchange_mode = "none"
        if self.standin > 1:
            self.exchange_mode = "captured"
        elif self.world > 1:
            backend = torch.distributed.get_backend(process_group)
            want = os.environ.get("TG_EXCHANGE", "captured" if backend == "nccl" else "eager")
            self.exchange_mode = "captured" if (want == "captured" and backend == "nccl") else "eager-split"
        self.B, self.T0, self.cs = F.batch_size, F.RNN_N, F.crop_size
        self.T = 2 * self.T0 - 1 if F.pingpang else self.T0
        self.use_vgg = F.vgg_scaling > 0
        specs = OrderedDict()
        specs["generator"] = generator_spec(F.num_resblock)
        specs["fnet"] = fnet_spec()
        # Dt_mergeDs=False: temporal-only D on the 9 warped channels, centre-cropped (lib/Teco.py:246-250,269-272)
        self.d_cin = 27 if F.Dt_mergeDs else 9
        if gan:
            specs["tdiscriminator"] = discriminator_spec(self.d_cin)
        self.ps = ParamStore(specs, self.dev, act_dtype,
                             bpad=("generator/generator_unit/output_stage/conv/Conv/weights",
                                   "fnet/autoencode_unit/output_stage/conv2/Conv/weights"))
        vals = OrderedDict()
        vals.update(init_values(specs["generator"], seed))
        vals.update(init_values(specs["fnet"], seed + 1))
        if gan:
            vals.update(init_values(specs["tdiscriminator"], seed + 2))
        self.ps.load(vals)
        self.G, self.Fn = Generator(self.ps, F.num_resblock), FNet(self.ps)
        self.D = Discriminator(self.ps) if gan else None
        if self.use_vgg:
            self.vps = ParamStore(OrderedDict(vgg=vgg_spec()), self.dev, act_dtype, trainable=False, wide_frag=True)
            self.vps.load(init_values(vgg_spec(), seed + 3, he_normal=True))
            self.V = VGG19(self.vps)
        # optimisers: index 0 = discriminator (gated) when present
        self.opt_scopes = (["tdiscriminator"] if gan else []) + ["generator", "fnet"]
        nopt = len(self.opt_scopes)
        st = torch.zeros(8 + 2 * nopt, dtype=torch.float64)
        st[2], st[3], st[4], st[5] = F.learning_rate, F.decay_step, F.decay_rate, 1.0 if F.stair else 0.0
        st[6] = F.Dbalance
        st[7] = 1.0 if F.Dt_mergeDs else 0.3                     # lib/Teco.py:423-424
        self.sched = st.to(self.dev)
        self.hyper = torch.zeros(nopt, 8, device=self.dev)
        # one fp32 scratch, zeroed by ONE fill per step: the loss slots, then the batch-norm statistics / backward sums
        # of the 2 forward + 3 backward discriminator passes
        # (the two forward passes keep K.BN_STAT_REPLICAS partial sets each: csrc/conv4x4s2.hip's epilogue)
        nbn = (3 + 2 * K.BN_STAT_REPLICAS) * 2 * sum(co for _, _, co in DIS_BLOCKS) if gan else 0
        self.zbuf = torch.zeros(len(LOSS_NAMES) + nbn, device=self.dev)
        self.loss = self.zbuf[:len(LOSS_NAMES)]
        self.bn_pool = self.zbuf[len(LOSS_NAMES):]
        h = self.cs
        self.in_lr = torch.zeros(self.B, self.T0, h, h, 3, device=self.dev)
        self.in_hr = torch.zeros(self.B, self.T0, 4 * h, 4 * h, 3, device=self.dev)
        self.seq_idx = list(range(self.T0)) + (list(range(self.T0 - 2, -1, -1)) if F.pingpang else [])
        self._segs = None
        self.lazy_side = True             # just-in-time side-stream launches (segments.plan_launch_order; False: +1.4 ms, lesson 7)
        # measurement mode: device wall-clock stamps at every segment boundary (one-thread kernels, captured with the segment)
        self.seg_stamps = torch.zeros(128, dtype=torch.int64, device=device) if os.environ.get("TG_SEG_STAMPS") else None
        self.seg_stamp_names = {}
        self._pools = {}
        self._done, self._mode, self._main, self._d_vgg = {}, "flat", None, None
        # the fade-in factor of the adversarial / layer losses (lib/Teco.py:379-380) is a DEVICE scalar derived from the
        # device-side step counter at the head of every step: Dt_ratio_add != 0 stays inside the captured graph
        self.dt_ratio = torch.ones(1, device=self.dev)
        self.use_graph = use_graph
        self.host_step = 0
        self._skip_update = False
        self.gen = None
        self.side_stream = _shared_stream(self.dev, "S")
        # two-stream overlap of the latency-bound chain with throughput work (see _program_compute); TG_OVERLAP=0: A/B
        self.overlap = os.environ.get("TG_OVERLAP", "1") != "0"
        # which pieces go to the side stream (A/B bit mask): 1 VGG target features, 2 D real pass, 4 VGG pass of the early
        # frames, 8 D's own-gradient passes, 32 the generator's weight gradients beside FNet's backward pass, 64 the VGG pass
        # of the late frames beside D's generator-side backward pass (before the BPTT)
        # (round 4: default 103, i.e. bit 8 off -- D's own-gradient passes on the MAIN stream, in the ~1.1 ms it would otherwise
        #  wait for the last chunk's VGG pass, instead of beside the BPTT: 9.15 / 9.18 -> 9.07 / 9.09 ms, profiles/r04k_ab.txt)
        # (round 6: default 111, bit 8 ON again -- with the one-launch trunk the side stream's VGG passes are the critical path of the
        #  middle window and D's own-gradient passes beside the last of them cost it 0.3 ms (vgg_1 2.00 -> 1.70 ms without them);
        #  beside the BPTT they cost the BPTT 0.44 ms but the side stream has the room: 7.679 / 7.670 -> 7.608 / 7.601 ms and, another
        #  box, 7.485 / 7.480 -> 7.432 / 7.448; behind the target lookahead instead of in front of it: 7.455; profiles/r06aa_ab.txt)
        self.ov_parts = (int(os.environ.get("TG_OVERLAP_PARTS", "111")) & 111) if self.overlap else 0
        self.wgrad_cut = 0                       # BPTT cut for early generator weight gradients (see _program_compute); 0 = off
        # Ping-pong sequences repeat their first T0-1 TARGET frames in reverse (lib/Teco.py:80-85), so the VGG features of the
        # targets (lib/Teco.py:174-176) need computing for the T0 distinct frames only; the mirrored ones are copies.  Measured in
        # round 4 (same box, profiles/r04a_ab.txt): 10.97 -> 10.66 ms per TecoGAN step; TG_VGGT_DEDUP=0 is the A/B switch.
        self.vggt_dedup = os.environ.get("TG_VGGT_DEDUP", "1") == "1" and bool(F.pingpang) and self.T0 > 1
        # Target LOOKAHEAD (round 4).  The target features depend on the data only, and before the BPTT the side stream is the
        # step's critical path (target pass 1.4 ms + D real pass + 4 ms of VGG passes over the generated frames,
        # profiles/r04c_ab.txt) while it idles ~2 ms during the BPTT.  A caller that knows the NEXT batch's targets
        # (`step(x, y, next_targets=...)`: the loaders prefetch anyway) gets them put through VGG-19 during THIS step's BPTT
        # phase (segment `vggt_next`); the next step then starts from the stored features (segment `vggt_pre`: one gather).
        # Every step still computes one batch of target features -- one step earlier.  Without `next_targets` the features are
        # computed in-step as before (segment `vggt`).  A caller that never announces anything never uses the mechanism.
        self.lookahead = self.use_vgg
        self.in_hr_next = torch.zeros(self.B, self.T0, 4 * h, 4 * h, 3, device=self.dev) if self.lookahead else None
        self._taps_t, self._taps_next = None, None      # persistent feature buffers (outside the graph pools: they cross steps)
        self._next_ready = False                        # _taps_next holds the features of the batch the NEXT step() will get
        self._have_next = False                         # this step() was given next_targets
        self._announced = None                          # (tensor, its _version) announced as next_targets: identity check in step()
        self._hold = []
        self.comm_stream = _shared_stream(self.dev, "C") if self.world > 1 else None
        # Captured exchange segments can be launched from their own host thread (segments.SegmentRunner), so that the just-in-time
        # wait in front of a collective does not hold up the other streams' launches.  That thread has only ever launched the
        # stand-in kernels (no multi-GPU node was available to the builder), so with a REAL process group the default is the
        # caller's thread until a 2+ rank RCCL run has exercised it; TG_COMM_THREAD=1 / 0 decides explicitly.
        env = os.environ.get("TG_COMM_THREAD")
        self.comm_thread = (env != "0") if env is not None else (self.pg is None)
        # a captured exchange segment without a kernel node is an error whenever real ranks depend on it
        self.require_exchange_nodes = self.pg is not None and self.world > 1
        self.exchange_nodes = {}                 # segment -> (nodes, kernel nodes) of its captured graph
        self.exchange_segments = []              # names of the communication-stream segments of the captured program
        self.streams = {"S": self.side_stream, "C": self.comm_stream}
        # a step that uses a second stream (overlap pieces, RCCL) is replayed as a DAG of single-stream graph segments
        uses_side = (self.use_vgg and self.ov_parts & 69) or (gan and self.ov_parts & 10) or bool(self.ov_parts & 32)
        self.segmented = bool(uses_side) or self.world > 1
        # the chain's own launches also take co-residency-friendly tiles when something runs beside them: the HR deconv
        # (56 KB LDS) and the output conv (67 KB) would otherwise not fit next to a resident <8,64> VGG workgroup (109 KB)
        # and each such node would wait for a CU to drain (~50 us, 27 + 19 nodes per step)
        if uses_side:
            self.G.chain_flags = K.CONV_COEXIST
        self.lookahead = self.lookahead and self.segmented and bool(self.ov_parts & 1)

    # ------------------------------------------------------------------------------------------
    def set_batch(self, r_inputs, r_targets):
        """r_inputs [B,T0,h,w,3] in [0,1]; r_targets [B,T0,4h,4w,3] in [-1,1] (lib/Teco.py:78)."""
        self.in_lr.copy_(r_inputs, non_blocking=True)
        self.in_hr.copy_(r_targets, non_blocking=True)
        self._next_ready = False          # a new batch: stored target features (if any) belonged to the announced one only

    def step(self, r_inputs=None, r_targets=None, next_targets=None):
        """One training step.  next_targets (optional, [B,T0,4h,4w,3]): the targets of the batch the NEXT call will train on
        -- their VGG features are computed during this step's backward phase (target lookahead, see __init__); pass the
        very tensor (or a view of the same memory) the next call passes as r_targets, unmodified (next_targets=True: the
        resident batch stays, as in bench.py); a next call with any other tensor computes its target features in-step."""
        ready = self._next_ready
        if r_inputs is not None:
            # The previous call announced a batch and its target features are stored.  They are used only if THIS call's
            # r_targets is provably the announced tensor: the same memory, not modified in place since (tecogan_amd/promise.py)
            # -- no device sync, no silent substitution of the caller's targets.  Anything else (a skipped
            # or reshuffled batch, a resume, a fresh `.cuda()` copy of equal values) falls back to the in-step target pass.
            is_kept = ready and promise.kept(self._announced, r_targets)
            self.set_batch(r_inputs, r_targets)                  # (clears _next_ready)
            self._next_ready = is_kept
        elif ready and self._announced is not None:
            self._next_ready = False                             # a resident-batch step after a tensor announcement: not that batch
        self._have_next = self.lookahead and next_targets is not None
        self._announced = None
        if self._have_next:
            if next_targets is True:                             # the resident batch is also the next one (bench.py)
                self.in_hr_next.copy_(self.in_hr, non_blocking=True)
            else:
                self.in_hr_next.copy_(next_targets, non_blocking=True)
                self._announced = promise.announce(next_targets)
        self.host_step += 1
        self.used_stored_targets = bool(self._next_ready)      # (observable for tests / logs: this step starts from looked-ahead features)
        if not self.use_graph:
            self._run_program("eager")
        else:
            if self._segs is None:
                self._capture()
            self._replay()
        self._next_ready = self._have_next

    def eval_losses(self, r_inputs, r_targets):
        """The loss scalars of `losses()` on a batch WITHOUT updating anything that training reads (reference main.py:391-402:
        the validation fetches every summary_freq steps).  Runs the step's program eagerly and skips the update segment:
        weights, Adam moments, the step counter and the balance average are untouched (the gradient buffer is overwritten --
        every training step clears it first -- and D's unused batch-norm moving statistics take one more update)."""
        ready, have = self._next_ready, self._have_next
        self.set_batch(r_inputs, r_targets)
        self._skip_update, self._have_next = True, False        # (validation data: features in-step, none stored)
        try:
            self._run_program("eager")
        finally:
            self._skip_update = False
        self._next_ready, self._have_next = ready, have         # the training stream's stored features are untouched
        torch.cuda.synchronize(self.dev)
        return self.losses()

    # ------------------------------------------------------------------------------------------
    def _program(self):
        self._program_compute()
        if self.gan and self._mode in ("eager", "capture", "flat"):
            # the BN scratch pool is valid inside a step's program only (captured graphs hold the pool's addresses, not this
            # attribute): D.forward / D.backward on eng.D outside a step get self-zeroed tensors again (sticky cursor)
            self.D.set_scratch(None)
        if self._skip_update:
            return
        # (vggt_next reads in_hr_next and writes _taps_next on the side stream: the next step()'s host-side copies into those
        #  buffers are ordered on the MAIN stream, so the update segment joins it explicitly -- ADVICE r4: without overlap bit
        #  32 nothing else did)
        after = ["down", "wgrad", "vggt_next"]
        if self.exchange_mode == "eager-split":
            self._seg_call("exchange", "M", after, self._allreduce)
            after = ["exchange"]
        elif self.exchange_mode == "captured":
            after = ["down", "wgrad", "vggt_next", "ar_d", "ar_g", "ar_f"]
        with self._seg("update", "M", after):
            self._program_update()

    def _program_compute(self):
        """Forward + backward of one step as segments on two streams.

        The recurrent generator chain (T frames x ~38 dependent launches forward, as many backward) is latency bound:
        each launch keeps one wave per SIMD busy for ~3.6 us, a third of the step during which >90 % of the chip's issue
        slots idle.  Everything that does not depend on the chain's current position runs on the SIDE stream meanwhile,
        with tile shapes whose LDS / register footprint lets a chain workgroup co-reside on the same CU
        (TG_CONV_COEXIST; profiles/r02a_overlap.txt, r02d_forktax.txt):
            forward chain   ||  VGG target features, D real pass, VGG gen pass (fwd + dX) of the early frames
            backward chain  ||  D's own-gradient passes [, generator weight gradients of the late frames: measured a loss]
        TG_OVERLAP=0 runs the identical program on one stream (A/B switch; the results are the same either way)."""
        F, ps, B, T, h = self.F, self.ps, self.B, self.T, self.cs
        H = 4 * h
        hold = self._hold = []                  # tensors that cross segments / streams stay referenced until the next step
        seg = self._seg

        def part(bit):
            """(stream key, conv footprint flag) of schedule piece `bit`."""
            return ("S", K.CONV_COEXIST) if (self.ov_parts & bit) else ("M", 0)

        # (starting the target-frame VGG pass already beside FNet's forward pass -- a separate "prep" segment for the fills and
        #  gathers -- measured a LOSS: 12.59 / 12.66 -> 12.89 / 12.92 ms, profiles/r03o_ab.txt)
        with seg("head"):
            ps.grad.zero_()
            self.zbuf.zero_()
            if self.gan:
                self.D.set_scratch(self.bn_pool)
                K.dt_ratio(self.sched, F.Dt_ratio_0, F.Dt_ratio_add, F.Dt_ratio_max, self.dt_ratio)
            # ping-pong extension (lib/Teco.py:80-85) + [B,T] -> frame-major [T,B] in one gather each
            lr_seq = K.seq_gather(self.in_lr, torch.empty(T, B, h, h, 3, device=self.dev), self.seq_idx)
            hr_seq = K.seq_gather(self.in_hr, torch.empty(T, B, H, H, 3, device=self.dev), self.seq_idx)
            npair = (T - 1) * B
            # ---- FNet on all consecutive pairs (lib/Teco.py:102-117) ------------------------------------
            pre_lr = lr_seq[:-1].reshape(npair, h, h, 3)
            cur_lr = lr_seq[1:].reshape(npair, h, h, 3)
            fnet_in = K.concat2_pad(pre_lr, cur_lr,
                                    torch.empty(npair, h, h, FNET_CPAD, device=self.dev, dtype=self.act_dtype))
            flow, fsaved = self.Fn.forward(fnet_in)                                              # [npair,h,h,2] fp32
            flow_t = flow.view(T - 1, B, h, h, 2)
            gd = self._gan_setup(lr_seq, flow_t) if self.gan else None
            if self.gan:
                # the real and the fake pass write the two halves of ONE set of [2 tb, ...] activation buffers: D's own backward
                # pass then runs once over both (Discriminator.backward_pair)
                gd["pair"] = self.D.alloc_pair(gd["tb"], gd["Ho"], gd["Ho"], lr_seq)
            # ---- LR warp loss (lib/Teco.py:120-122,329-333) and its gradient to the flow ----------------
            warped_lr = K.warp_forward(pre_lr, flow, torch.empty_like(pre_lr))
            npx = float(npair * h * h)
            K.sum_sq_diff(cur_lr, warped_lr, 1.0 / npx, self.loss[LI["l2_warp_loss"]:LI["l2_warp_loss"] + 1])
            c = 2.0 * F.warp_scaling / npx
            d_wl = K.lincomb(warped_lr, cur_lr, torch.empty_like(warped_lr), c, -c)
            d_flow = torch.empty_like(flow)
            K.warp_backward(d_wl, pre_lr, flow, None, d_flow)
        hold += [lr_seq, hr_seq, fnet_in, flow, fsaved, gd, warped_lr, d_wl, d_flow]
        # ---- side: VGG-19 features of the targets (lib/Teco.py:177-178; needs hr_seq only) and the discriminator on the
        #      real triplets (needs the flows, not the generator) -- beside the first half of the forward chain
        taps_t = None
        if self.use_vgg:
            sk, cx = part(1)
            Tu = self.T0 if self.vggt_dedup else T                      # frames whose features are actually computed
            taps_t = self._alloc_taps(Tu)
            in_step = (lambda: not self._next_ready) if self.lookahead else None
            if in_step is None or self._seg_on("vggt", sk, in_step):
                with seg("vggt", sk, ["head"], cond=in_step):
                    xt = K.vgg_preprocess_forward(hr_seq[:Tu].view(Tu * B, H, H, 3),
                                                  torch.empty(Tu * B, H, H, VGG_CPAD, device=self.dev, dtype=self.act_dtype))
                    taps_u, _ = self.V.forward(xt, keep=False, flags=cx)
                    self._spread_taps(taps_u, Tu)                       # frame-major [Tu*B,...] -> [T*B,...] in sequence order
                    hold += [xt, taps_u]
            if self.lookahead and self._seg_on("vggt_pre", sk, lambda: self._next_ready):
                # the features were computed during the previous step's backward phase (segment vggt_next of that step)
                with seg("vggt_pre", sk, ["head"], cond=lambda: self._next_ready):
                    self._spread_taps(self._taps_next, Tu)
        if self.gan:
            sk, cx = part(2)
            with seg("dreal", sk, ["head"]):
                into = self.D.pair_half(gd["pair"], 0)
                gd["real"] = K.pack_d_input_forward(hr_seq, lr_seq, *gd["args"], into["x"], B, h, h, gd["off"], gd["merge"])
                gd["p_real"], gd["l_real"], gd["sv_real"] = self.D.forward(gd["real"], flags=cx, into=into)
        # ---- recurrent generator (lib/Teco.py:125-155) in CHUNKS of frames; side: the VGG pass (forward, cosine loss against the
        #      target features, input gradient) of every chunk as soon as its frames exist, beside the forward recurrence of
        #      the next chunk -- the last one beside the loss / D work.
        # The cuts are `_vgg_cuts(T)` (TG_VGG_CUTS overrides): one cut a little past the middle; more, smaller chunks start the
        # passes earlier but lose more to the fixed cost of a pass than they gain (round 3: profiles/r03x_ab.txt; round 4 with the
        # shorter chain and the target lookahead: profiles/r04f_ab.txt).
        # (Schedules measured and rejected in round 3, profiles/r03c_ab.txt, r03e_ab.txt: VGG passes of frame-DESCENDING pieces
        #  beside the BPTT of the frames above them, FNet's backward pass / the generator's weight gradients of the late frames
        #  beside the BPTT of the early ones.  The last chunk's pass on the side stream beside D's generator-side backward pass
        #  on the main stream was that round's one gain: 12.87 -> 12.68 ms.)
        gen = torch.empty(T, B, H, H, 3, device=self.dev) if self.gen is None else self.gen
        self.gen = gen
        cuts = [0] + (self._vgg_cuts(T) if self.use_vgg else []) + [T]
        chunks = [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
        d_vgg = None
        if self.use_vgg:
            d_vgg = self._d_vgg = (torch.empty(T, B, H, H, 3, device=self.dev) if self._d_vgg is None else self._d_vgg)

        def forward_frames(t0, t1):
            for t in range(t0, t1):
                K.warp_s2d_forward(gen[t - 1] if t else None, flow_t[t - 1] if t else None, lr_seq[t], self.G.seq["x_in"][t],
                                   0.5, 0.5)
                self.G.forward_t(t, gen[t])

        split = self._mode != "flat"
        vgg_segs = []
        for k, (t0, t1) in enumerate(chunks):
            fname = "fwd_%d" % k
            with seg(fname):
                if k == 0 and (self.G.seq is None or self._mode != "capture"):
                    self.G.begin_sequence(T, B, h, h, self.dev)
                forward_frames(t0, t1)
            if self.use_vgg:
                last = k == len(chunks) - 1 and k > 0
                # (bit 4: the chunks that run beside the forward chain, with co-residency-friendly tiles; bit 64: the last chunk,
                #  beside the loss / D work -- throughput beside throughput, full tiles)
                on_side = split and bool(self.ov_parts & (64 if last else 4))
                with seg("vgg_%d" % k, "S" if on_side else "M", [fname, "vggt", "vggt_pre"]):
                    self._vgg_chunk(gen, taps_t, t0, t1, d_vgg, K.CONV_COEXIST if (on_side and not last) else 0, zero=True)
                vgg_segs.append("vgg_%d" % k)

        def losses_and_fake_pass():
            # ---- generator losses seeded into d_gen -------------------------------------------------------
            nhr = float(T * B * H * H)
            K.sum_sq_diff(gen, hr_seq, 1.0 / nhr, self.loss[LI["l2_content_loss"]:LI["l2_content_loss"] + 1])
            d_gen = K.lincomb(gen, hr_seq, torch.empty_like(gen), 2.0 / nhr, -2.0 / nhr)         # lib/Teco.py:320-322
            if F.pingpang:
                self._pingpong(gen, d_gen)
            if self.gan:
                into = self.D.pair_half(gd["pair"], 1)
                gd["fake"] = K.pack_d_input_forward(gen, lr_seq, *gd["args"], into["x"], B, h, h, gd["off"], gd["merge"])
                gd["p_fake"], gd["l_fake"], gd["sv_fake"] = self.D.forward(gd["fake"], into=into)
                self._gan_losses(gd)
            return d_gen

        with seg("fwd_loss", "M", ["dreal", "vggt", "vggt_pre"]):
            d_gen = losses_and_fake_pass()
        hold.append(d_gen)
        fwd_last = "fwd_loss"                                    # the segment D's passes and the losses are complete in
        if self.gan:
            sk, cx = part(8)
            with seg("down", sk, [fwd_last]):     # D's own gradients (t_discrim_loss) from both passes, one sweep over 2 tb samples
                self.D.backward_pair(gd["pair"], (gd["sv_real"], gd["sv_fake"]), gd["d_own_D"], flags=cx)
            # D's gradients and t_balance are final: their all-reduce overlaps the rest of the backward pass
            self._exchange_seg("ar_d", ["tdiscriminator"], ["down"], with_balance=True)
        if self.lookahead and not self._skip_update and self._seg_on("vggt_next", "S", lambda: self._have_next):
            # target lookahead: the NEXT batch's target features, beside this step's BPTT (the side stream idles there)
            with seg("vggt_next", "S", [fwd_last], cond=lambda: self._have_next):
                Tu = self.T0 if self.vggt_dedup else T
                hn = K.seq_gather(self.in_hr_next, torch.empty(Tu, B, H, H, 3, device=self.dev), self.seq_idx[:Tu])
                xn = K.vgg_preprocess_forward(hn.view(Tu * B, H, H, 3),
                                              torch.empty(Tu * B, H, H, VGG_CPAD, device=self.dev, dtype=self.act_dtype))
                taps_n, _ = self.V.forward(xn, keep=False, flags=K.CONV_COEXIST)
                for key, u in taps_n.items():
                    self._taps_next[key].copy_(u)
                hold += [hn, xn, taps_n]
        # ---- backward through the recurrence ------------------------------------------------------------------
        d_flow_t = d_flow.view(T - 1, B, h, h, 2)
        tail_split = self.exchange_mode == "captured"    # the RCCL segments hook in after wgrad and after FNet's backward
        gw_side = bool(self.ov_parts & 32) and split

        def backward_frames(t1, t0):
            for t in range(t1 - 1, t0 - 1, -1):
                dx_in = self.G.backward_t(t, d_gen[t], need_dx=t > 0)
                if t > 0:
                    K.warp_s2d_backward(dx_in, gen[t - 1], flow_t[t - 1], d_gen[t - 1], d_flow_t[t - 1], 0.5)

        if self.gan:
            with seg("bwd", "M", []):
                # generator-side gradient through the fake pass (adversarial + layer loss): no D weight gradients
                dx = self.D.backward(gd["sv_fake"], gd["d_fake_G"], gd["d_layers"], wgrad=False, need_dx=True)
                K.pack_d_input_backward(dx, gen, gd["args"][0], gd["args"][1], gd["args"][2], gd["args"][3], d_gen, B, h, h,
                                        gd["off"], gd["merge"])
                hold.append(dx)
        # BPTT cut (round 6, `wgrad_cut` = c > 0): the generator's weight gradients of frames c .. T-1 on the side stream beside
        # the BPTT of frames c-1 .. 0 -- the side stream idles there once the target lookahead is through -- so that the tail
        # beside FNet's backward pass carries the weight gradients of c frames only.  (Round 4 measured this a loss with the
        # per-block chain, profiles/r04u_ab.txt; re-measured with the one-launch trunk: profiles/r06w_ab.txt.)
        wcut = self.wgrad_cut if (gw_side and 0 < self.wgrad_cut < T) else 0
        with seg("bwd_b", "M", vgg_segs):
            if self.use_vgg:
                K.lincomb(d_vgg, None, d_gen, 1.0, 0.0, accumulate=True)           # the chunks' perceptual-loss gradients
            backward_frames(T, wcut)
            if not tail_split and not gw_side:
                self.G.wgrad_sequence(0, T)
                self.Fn.backward(fsaved, d_flow)
        if wcut:
            with seg("wgrad_a", "S", ["bwd_b", "vggt_next"]):
                self.G.wgrad_sequence(wcut, T, flags=K.CONV_COEXIST)
            with seg("bwd_c", "M", []):
                backward_frames(wcut, 0)
        if tail_split or gw_side:
            with seg("wgrad", "S" if gw_side else "M", ["bwd_c" if wcut else "bwd_b", "wgrad_a"]):       # (bit 32: beside FNet's backward pass)
                self.G.wgrad_sequence(0, wcut if wcut else T)
            self._exchange_seg("ar_g", ["generator"], ["wgrad"])        # overlaps the FNet backward pass
            # (FNet's 14 weight gradients behind the generator's on the side stream, its input-gradient chain alone on the
            #  main stream: measured no gain -- 12.69 vs 12.73 ms, profiles/r03j_ab.txt -- the pieces serialise on each other.
            #  Round 4, commits f848561 / 8541e73 in the history: the BPTT cut at frame k, with FNet's backward pass of the pairs
            #  above k and / or the generator's weight gradients of the frames above k on the idle side stream beside the BPTT of
            #  the frames below -- FNet slice neutral (8.50 vs 8.50 ms: the pass is a latency-bound chain of 22 launches whose
            #  length does not depend on the batch, 354 us alone, 0.8-1.0 ms beside the weight gradients), early weight
            #  gradients a loss (9.20 -> 9.28-9.31 ms, FRVSR 2.50 -> 2.54-2.56: what they take from the BPTT beside them is
            #  more than what they leave to FNet's pass; profiles/r04s_ab.txt, r04u_ab.txt).  Deleted.)
            with seg("fnet_bwd"):
                self.Fn.backward(fsaved, d_flow)
            self._exchange_seg("ar_f", ["fnet"], ["fnet_bwd"])

    @staticmethod
    def _vgg_cuts(T):
        """Frame indices at which the forward recurrence is cut into chunks for the VGG passes (see _program_compute)."""
        env = os.environ.get("TG_VGG_CUTS")
        if env is not None:
            return sorted({int(c) for c in env.split(",") if c.strip() and 0 < int(c) < T})
        # ONE cut, a little before the middle (19 frames: 7 + 12).  Round 4 re-measured the alternatives with the shorter chain and
        # the target lookahead.  Several chunks (profiles/r04f_ab.txt, one box): 11 -> 9.19 ms; 6,12 -> 9.28; 7,14 -> 9.45;
        # 8,14 -> 9.53; 5,10,15 -> 10.03; 4,8,12,16 -> 10.48: a VGG pass has ~0.5 ms of fixed cost (60 launches, half of them
        # latency-bound; 1.26 / 1.41 / 1.84 / 2.64 ms alone for 20 / 32 / 44 / 76 images, profiles/r04g_ab.txt) and its small
        # layers need >= 32 images to fill the chip.  Where the one cut goes (r04g_ab.txt, another box): 6 -> 9.77 ms, 7 -> 9.73,
        # 8 -> 9.68, 9 -> 9.92, 11 -> 9.87 / 9.89, 13 -> 10.02: the side stream's VGG passes are the critical path up to the
        # BPTT, so the first chunk should exist early.  Re-measured on the round's final kernels (the forward chain another
        # 0.3 ms shorter; profiles/r04w_ab.txt, one box): 6 -> 8.50 ms, 7 -> 8.45 / 8.44, 8 -> 8.48 / 8.59 / 8.49 / 8.51, 9 -> 8.77.
        return [min(max(int(0.37 * T + 0.5), 1), T - 1)] if T > 1 else []

    def _alloc_taps(self, Tu):
        """Persistent target-feature buffers: `_taps_t` [T*B,...] (what the VGG passes over the generated frames compare with)
        and, with the lookahead, `_taps_next` [Tu*B,...] (the next batch's).  They cross steps, so they live outside the graph
        pools (allocated in the eager warm-up run, before any capture)."""
        if self._taps_t is None:
            H, n = 4 * self.cs, self.T * self.B
            dims = {VGG_TAPS[0]: (H // 2, 128), VGG_TAPS[1]: (H // 4, 256), VGG_TAPS[2]: (H // 8, 512), VGG_TAPS[3]: (H // 16, 512)}
            mk = lambda m: {k: torch.empty(m, hw, hw, c, device=self.dev, dtype=self.act_dtype) for k, (hw, c) in dims.items()}   # noqa: E731
            self._taps_t = mk(n)
            if self.lookahead:
                self._taps_next = mk(Tu * self.B)
        return self._taps_t

    def _spread_taps(self, taps_u, Tu):
        """[Tu*B,...] features of the distinct target frames -> `_taps_t` [T*B,...] in sequence order (ping-pong mirror)."""
        T = self.T
        for key, u in taps_u.items():
            full = self._taps_t[key]
            if Tu == T:
                full.copy_(u)
            else:
                K.seq_gather(u.view(1, Tu, -1).view(torch.float32), full.view(T, 1, -1).view(torch.float32), self.seq_idx)

    def _program_update(self):
        """Device-side schedule, the TF-Adams (D gated) and the refresh of the MFMA weight copies."""
        F, ps = self.F, self.ps
        tb = self.loss[LI["t_balance"]:LI["t_balance"] + 1] if self.gan else None
        K.schedule_step(self.sched, self.hyper, len(self.opt_scopes), 0 if self.gan else -1, tb, F.beta, 0.999,
                        F.adameps)
        for k, scope in enumerate(self.opt_scopes):
            a, b = ps.scope_range[scope]
            K.adam_tf(ps.flat[a:b], ps.grad[a:b], ps.m[a:b], ps.v[a:b], self.hyper[k], 1.0 / self.world)
        ps.repack()

    # ------------------------------------------------------------------------------------------
    def _slot(self, name):
        return self.loss[LI[name]:LI[name] + 1]

    def _pingpong(self, gen, d_gen):
        """lib/Teco.py:362-372: mean |gen[k] - gen[T-1-k]|, k < RNN_N-1, weighted by pp_scaling."""
        npair = self.T0 - 1
        cnt = float(npair * gen[0].numel())
        K.pingpong(gen, d_gen, self.T, npair, 1.0 / cnt, (self.F.pp_scaling if self.F.pp_scaling > 0 else 0.0) / cnt,
                   self._slot("PingPang"))

    def _vgg_chunk(self, gen, taps_t, t0, t1, dst, flags, zero):
        """lib/Teco.py:174-178,339-359 for frames [t0, t1): VGG-19 forward of the generated frames, cosine distance of the
        four L2-normalised taps against the target taps (means over ALL T*B frames: the scales use the full count), dX
        backward (weights frozen) accumulated into dst[t0:t1] (zero=True: dst is a scratch tensor, cleared first).  The loss slots hold mean cos; losses() reports 1 - cos."""
        F, T, B, H = self.F, self.T, self.B, 4 * self.cs
        n = (t1 - t0) * B
        xg = K.vgg_preprocess_forward(gen[t0:t1].view(n, H, H, 3),
                                      torch.empty(n, H, H, VGG_CPAD, device=self.dev, dtype=self.act_dtype))
        taps_g, acts = self.V.forward(xg, flags=flags)
        d_taps = {}
        for i, key in enumerate(VGG_TAPS):
            g = taps_g[key]
            t = taps_t[key][t0 * B:t1 * B]
            npix = float(T * B * g.shape[1] * g.shape[2])
            d = torch.empty_like(g)
            K.cosine_loss(g, t, 1.0 / npix, -F.vgg_scaling / npix, self._slot("vgg_loss_%d" % (i + 2)), d)
            d_taps[key] = d
        dx = self.V.backward(acts, d_taps, flags=flags)
        dv = dst[t0:t1]
        if zero:
            dv.zero_()
        K.vgg_preprocess_backward(dx, dv)
        self._hold += [xg, taps_g, acts, d_taps, dx]

    def _gan_setup(self, lr_seq, flow_t):
        """Triplet bookkeeping of lib/Teco.py:180-220."""
        F, T, B, h = self.F, self.T, self.B, self.cs
        H = 4 * h
        t_size = 3 * (T // 3)
        nt = t_size // 3
        tb = B * nt
        off = 0
        if F.crop_dt < 1.0:                                           # lib/Teco.py:216-220
            off = (H - int(H * F.crop_dt)) // 2
        idx_pre = list(range(0, t_size, 3))                           # forward motion re-used (Teco.py:201,207)
        if F.pingpang:
            idx_nxt = list(range(T - 1))[-2:-1 - t_size:-3]           # backward motion re-used (Teco.py:209)
            flow_nxt = flow_t
        else:                                                         # backward motion from FNet (Teco.py:190-199)
            back_in = K.concat2_pad(lr_seq[2:t_size:3].reshape(tb, h, h, 3), lr_seq[1:t_size:3].reshape(tb, h, h, 3),
                                    torch.empty(tb, h, h, FNET_CPAD, device=self.dev, dtype=self.act_dtype))
            flow_back, _ = self.Fn.forward(back_in, keep=False)       # stop_gradient (Teco.py:214)
            flow_nxt, idx_nxt = flow_back.view(nt, B, h, h, 2), list(range(nt))
        # merge: [before | warped (zero border) | bilinear LR context] = 27 channels at full size (lib/Teco.py:234-245);
        # otherwise only the 9 warped channels, cropped to (4h - 2 off)^2 (lib/Teco.py:231-232,249-250)
        merge = bool(F.Dt_mergeDs)
        return dict(tb=tb, off=off, merge=merge, Ho=H if merge else H - 2 * off, args=(flow_t, flow_nxt, idx_pre, idx_nxt),
                    flow_nxt=flow_nxt)

    def _gan_losses(self, gd):
        """lib/Teco.py:275-313,374-417: adversarial / discriminator / balance scalars and the layer losses, with the gradient
        seeds of the three D backward passes."""
        F = self.F
        p_real, l_real, p_fake, l_fake = gd["p_real"], gd["l_real"], gd["p_fake"], gd["l_fake"]
        # (the two seeds of D's own backward pass are the halves of one [2 tb, ...] tensor, in the order of the pair buffers)
        tb = p_real.shape[0]
        gd["d_own_D"] = torch.empty((2 * tb,) + tuple(p_real.shape[1:]), dtype=p_real.dtype, device=p_real.device)
        gd["d_real_D"], gd["d_fake_D"], gd["d_fake_G"] = gd["d_own_D"][:tb], gd["d_own_D"][tb:], torch.empty_like(p_real)
        # the five scalars land straight in their (contiguous) loss slots
        i0 = LI["t_adversarial_loss"]
        assert LOSS_NAMES[i0:i0 + 5] == ["t_adversarial_loss", "t_discrim_loss", "t_balance", "t_discrim_real_output",
                                         "t_discrim_fake_output"]
        K.gan_losses(p_real, p_fake, F.EPS, F.ratio, self.loss[i0:i0 + 5], gd["d_real_D"], gd["d_fake_D"],
                     gd["d_fake_G"], adv_scale_dev=self.dt_ratio)             # x dt_ratio (device scalar, Teco.py:379-384)
        gd["d_layers"] = None
        if F.D_LAYERLOSS:                                             # Teco.py:275-313,389-390
            gd["d_layers"] = []
            for i, norm in enumerate((12.0, 14.0, 24.0, 100.0)):
                r, f = l_real[i], l_fake[i]
                npix = float(r.numel() // r.shape[-1])
                d = torch.empty_like(f)
                K.l1_loss(r, f, 1.0 / npix, 0.02 / norm / npix, self._slot("D_layer_%d_loss" % i), d,
                          grad_scale_dev=self.dt_ratio)
                gd["d_layers"].append(d)

    # ------------------------------------------------------------------------------------------
    def losses(self, vec=None, one=1.0):
        """Loss values under the reference's names (lib/Teco.py update_list_name): of the last step, or -- with `vec` a
        zero-initialised 0.99-EMA of the raw loss slots and `one` = 1 - 0.99^updates -- the running averages the reference
        prints (update_list_avg: tf.train.ExponentialMovingAverage(0.99) over tensors starts from 0, no debiasing; every
        reported quantity is affine in the slots, `one` is the EMA of the constant 1 in `1 - mean cos`)."""
        F = self.F
        raw = OrderedDict(zip(LOSS_NAMES, (self.loss if vec is None else vec).detach().cpu().tolist()))
        out = OrderedDict()
        gen_loss = raw["l2_content_loss"]
        if self.gan and F.D_LAYERLOSS:
            s = 0.0
            for i, norm in enumerate((12.0, 14.0, 24.0, 100.0)):
                out["D_layer_%d_loss" % i] = raw["D_layer_%d_loss" % i]
                s += 0.02 * raw["D_layer_%d_loss" % i] / norm
            out["D_layer_loss_sum"] = s
        out["l2_content_loss"], out["l2_warp_loss"] = raw["l2_content_loss"], raw["l2_warp_loss"]
        if self.use_vgg:
            tot = 0.0
            for k in range(2, 6):
                out["vgg_loss_%d" % k] = one - raw["vgg_loss_%d" % k]      # slots hold mean cos
                tot += out["vgg_loss_%d" % k]
            out["vgg_all"] = tot
            gen_loss += F.vgg_scaling * tot
        if F.pingpang:
            out["PingPang"] = raw["PingPang"]
            if F.pp_scaling > 0:
                gen_loss += F.pp_scaling * raw["PingPang"]
        if self.gan:
            dt_ratio = float(self.dt_ratio.item())                          # the factor the last step ran with
            for k in ("t_adversarial_loss", "t_discrim_loss", "t_discrim_real_output", "t_discrim_fake_output"):
                out[k] = raw[k]
            gen_loss += F.ratio * raw["t_adversarial_loss"] * dt_ratio
            if F.D_LAYERLOSS:
                gen_loss += out["D_layer_loss_sum"] * dt_ratio
            out["t_balance_now"] = raw["t_balance"]
            out["t_balance"] = float(self.sched[1].item())                 # EMA (lib/Teco.py:415-417)
        out["All_loss_Gen"] = gen_loss
        return out

    def global_step(self):
        return int(self.sched[0].item())

    def check_handoffs(self):
        """Raise if a one-launch trunk lost a workgroup (see Generator.handoff_give_ups): called where the host synchronises anyway
        (checkpoints, the end of a bench run)."""
        n = self.G.handoff_give_ups()
        if n:
            raise RuntimeError("tg_resblock_chain: %d workgroup(s) gave up waiting for a neighbour (a launch did not get all its "
                               "workgroups resident): results since are invalid" % n)
