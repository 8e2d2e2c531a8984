"""Streaming 4x inference: the stateful per-frame recurrence of reference main.py:195-260 as one
hipGraph-captured kernel program per frame.

State (main.py:197-199): pre_inputs (previous LR frame), pre_gen (previous HR output in [0,1]); the
reference's third variable `pre_warp` (the warped HR frame) is never materialised here -- the fused
warp + space-to-depth kernel writes the generator input directly.  Per frame (main.py:201-216,253-260):
    flow   = fnet(concat(pre_inputs, frame))            (shrinks to a multiple of 8, SYMMETRIC-mirrored
                                                          back inside the warp kernel, main.py:188-190,212)
    x_in   = concat(frame, space_to_depth(dense_image_warp(pre_gen, upscale_four(4*flow)), 4))
    pre_gen = deprocess(generator_F(x_in))
The reference skips the FNet/warp on the first frame (pre_warp is still zero, main.py:257); running it is
equivalent because warping the all-zero initial pre_gen yields zeros, so one graph serves every frame.

Frame lookahead.  flow(t+1) depends on LR frames only (main.py:201-203: fnet sees pre_inputs and the new frame, never the HR
state), and the reference's loop holds the whole clip before it starts (lib/dataloader.py:30-60).  `step(frame, next_frame=...)`
therefore runs FNet for the NEXT frame on a second HIP stream beside this frame's generator (forked after the warp kernel, joined
at the end of the step, both inside the one captured graph): the 14 small FNet convs (0.22 of a 1.0 ms 1080p frame,
profiles/r03z_infer1080p_bf16_kernel_stats.txt) leave the frame's critical path.  The announced frame is a promise: the next call
must pass that very memory, unmodified (checked by tecogan_amd/promise.py: data pointer, layout and torch's version counter, no
sync; any other tensor makes the step compute its own flow first, as a call without `next_frame` does).

Lookahead WINDOW (round 6).  Alone, FNet on one 1080p frame pair is a chain of ~30 latency-bound launches: 0.31 ms for 32 GFLOP, and
since the generator's residual trunk became one persistent launch that owns every compute unit's LDS (csrc/resblock_plane.hip)
nothing runs beside it any more -- the side stream bought nothing (0.675 ms with, 0.672 without).  The flows of the next K frames
depend on LR frames only, so `step(frame, upcoming=[f1 .. fK])` runs FNet ONCE on the K pairs (frame, f1), (f1, f2), ... as a batch
when its stock of flows is used up (the same kernels at 8 x the pixels per launch: throughput instead of latency), and the following
K calls take their flow from the stock -- each against the same promise check as above: a frame that is not the announced memory
drops the whole stock and the step computes its own flow.  The reference's loop holds the whole clip (lib/dataloader.py:30-60,
main.py:253-260), so its driver can always announce.
"""
from collections import OrderedDict

import torch

from . import kernels as K
from . import promise
from .nets import FNET_CPAD, GEN_CPAD, FNet, Generator
from .params import ParamStore, fnet_spec, generator_spec, init_values
from .streams import capture_guard


class InferenceEngine:
    def __init__(self, num_resblock, h, w, device="cuda", act_dtype=torch.bfloat16, batch=1, seed=42, use_graph=True):
        self.dev, self.act_dtype, self.B, self.h, self.w = torch.device(device), act_dtype, batch, h, w
        specs = OrderedDict(generator=generator_spec(num_resblock), fnet=fnet_spec())
        self.ps = ParamStore(specs, self.dev, act_dtype, trainable=False)
        vals = OrderedDict()
        vals.update(init_values(specs["generator"], seed))
        vals.update(init_values(specs["fnet"], seed + 1))
        self.ps.load(vals)
        self.G, self.Fn = Generator(self.ps, num_resblock), FNet(self.ps)
        self.frame = torch.zeros(batch, h, w, 3, device=self.dev)                 # static input (placeholder)
        self.frame_next = torch.zeros(batch, h, w, 3, device=self.dev)            # the announced next frame
        self.pre_inputs = torch.zeros(batch, h, w, 3, device=self.dev)
        self.pre_gen = torch.zeros(batch, 4 * h, 4 * w, 3, device=self.dev)
        self.flow_next = torch.zeros(batch, h - h % 8, w - w % 8, 2, device=self.dev)   # flow(frame -> frame_next), side stream
        self.lookahead = True                                                      # False: next_frame ignored (FNet in line, one stream)
        self._announced = None                                                     # (tensor, its _version) passed as next_frame
        self._have_flow = False                                                    # flow_next belongs to the coming step
        self.side = torch.cuda.Stream(device=self.dev) if self.dev.type == "cuda" else None
        self.use_graph, self.graphs = use_graph, {}
        # lookahead window: FNet on the next `window` frame pairs as ONE batch (step(frame, upcoming=[...]))
        self.window = 16
        self._win_in = None                                                        # [window + 1, B, h, w, 3]: resident frame + announced ones
        self._win_flows = None                                                     # [window, B, h', w', 2]
        self._stock = []                                                           # [(promise, index into _win_flows)] in arrival order
        self._win_graphs = {}

    def load(self, values):
        """values: TF-variable-name -> tensor for the 'generator' and 'fnet' scopes (main.py:221-224)."""
        self.ps.load(values)

    def check_handoffs(self):
        """Raise if the one-launch trunk lost a workgroup (Generator.handoff_give_ups; synchronises): the end of a clip is the place."""
        n = self.G.handoff_give_ups()
        if n:
            raise RuntimeError("tg_resblock_plane: %d workgroup(s) gave up waiting for a neighbour (a launch did not get all its "
                               "workgroups resident): frames since are invalid" % n)

    def reset(self):
        self.pre_inputs.zero_()
        self.pre_gen.zero_()
        self._have_flow = False
        self._stock = []

    def _flow(self, prev, cur):
        B, h, w = self.B, self.h, self.w
        fin = K.concat2_pad(prev, cur, torch.empty(B, h, w, FNET_CPAD, device=self.dev, dtype=self.act_dtype))
        return self.Fn.forward(fin, keep=False)[0]                                 # [B, h-h%8, w-w%8, 2]

    def _program(self, have_flow, ahead):
        """have_flow: flow_next (computed beside the previous frame) is this frame's flow; ahead: compute the next one."""
        B, h, w = self.B, self.h, self.w
        flow = self.flow_next if have_flow else self._flow(self.pre_inputs, self.frame)
        x_in = torch.empty(B, h, w, GEN_CPAD, device=self.dev, dtype=self.act_dtype)
        K.warp_s2d_forward(self.pre_gen, flow, self.frame, x_in, 1.0, 0.0)        # state already in [0,1]
        main = torch.cuda.current_stream()
        if ahead:
            self.side.wait_stream(main)                                            # the warp has consumed flow_next
            with torch.cuda.stream(self.side):
                self.flow_next.copy_(self._flow(self.frame, self.frame_next))
        # generator; the fused bicubic / preprocess epilogue writes deprocess(frame) straight into the recurrent state
        # (the warp kernel above has consumed the old state by then: stream order)
        self.G.forward(x_in, keep=False, out=False, state=self.pre_gen)
        self.pre_inputs.copy_(self.frame)
        if ahead:
            main.wait_stream(self.side)

    def _window_flows(self, k):
        """FNet on the k pairs (win_in[i], win_in[i + 1]) as one batch -> win_flows[:k]."""
        B, h, w = self.B, self.h, self.w
        fin = K.concat2_pad(self._win_in[:k].reshape(k * B, h, w, 3), self._win_in[1:k + 1].reshape(k * B, h, w, 3),
                            torch.empty(k * B, h, w, FNET_CPAD, device=self.dev, dtype=self.act_dtype))
        flows = self.Fn.forward(fin, keep=False)[0]
        self._win_flows[:k].copy_(flows.view(k, B, *flows.shape[1:]))

    def _refill(self, upcoming):
        """The flows of the next frames (each against its predecessor; the first against the frame resident from this step)."""
        k = min(len(upcoming), self.window)
        if self._win_in is None:
            self._win_in = torch.zeros(self.window + 1, *self.frame.shape, device=self.dev)
            self._win_flows = torch.zeros(self.window, *self.flow_next.shape, device=self.dev)
        self._win_in[0].copy_(self.frame)
        for i in range(k):
            self._win_in[i + 1].copy_(upcoming[i], non_blocking=True)
        if not self.use_graph or k < self.window:
            self._window_flows(k)                                                  # (a clip's last, shorter window: once per clip, eager)
        else:
            if k not in self._win_graphs:
                self._window_flows(k)                                              # eager warm-up (allocations, weight copies)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # (thread-local capture mode: a caller's writer thread may be issuing its own copies meanwhile, main.py)
                with capture_guard(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._window_flows(k)
                self._win_graphs[k] = g
            self._win_graphs[k].replay()
        self._stock = [(promise.announce(upcoming[i]), i) for i in range(k)]

    def step(self, frame=None, next_frame=None, upcoming=None):
        """frame: [B,h,w,3] fp32 in [0,1] (device tensor).  next_frame: the frame the NEXT call will pass (its flow is computed
        beside this frame's generator).  upcoming: the frames the next calls will pass, in order (any number; the engine takes
        `window` of them whenever its stock of precomputed flows is used up and runs FNet on them as one batch); with `upcoming`
        the one-frame side-stream lookahead is not used.  Returns the HR frame [B,4h,4w,3] in [0,1] (a view of the recurrent
        state: copy it if you keep it across steps)."""
        if frame is not None:
            stocked = False
            if self._stock:
                if promise.kept(self._stock[0][0], frame):
                    self.flow_next.copy_(self._win_flows[self._stock.pop(0)[1]], non_blocking=True)
                    self._have_flow = stocked = True
                else:
                    self._stock = []                                               # not the announced frame: the stock is void
            # flow_next belongs to this frame only if it IS the announced tensor (same object, not written since): otherwise the
            # stored flow is dropped and the step computes its own (a caller that skips or reorders frames stays correct)
            if not stocked and self._have_flow and not promise.kept(self._announced, frame):
                self._have_flow = False
            self.frame.copy_(frame, non_blocking=True)
        else:
            self._stock = []
            if self._have_flow:
                self._have_flow = False                                            # re-running the resident frame: not the announced one
        windowed = bool(self._stock) or bool(upcoming)
        ahead = self.lookahead and next_frame is not None and not windowed
        self._announced = None
        if ahead:
            self.frame_next.copy_(next_frame, non_blocking=True)
            self._announced = promise.announce(next_frame)
        key = (self._have_flow, ahead)
        if not self.use_graph:
            self._program(*key)
        else:
            if key not in self.graphs:
                keep = (self.pre_inputs.clone(), self.pre_gen.clone(), self.flow_next.clone())
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self._program(*key)
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                self.pre_inputs.copy_(keep[0])
                self.pre_gen.copy_(keep[1])
                self.flow_next.copy_(keep[2])
                g = torch.cuda.CUDAGraph()
                with capture_guard(), torch.cuda.graph(g):
                    self._program(*key)
                self.graphs[key] = g
            self.graphs[key].replay()
        self._have_flow = ahead
        if upcoming and not self._stock and self.lookahead:
            self._refill(upcoming)
        return self.pre_gen
