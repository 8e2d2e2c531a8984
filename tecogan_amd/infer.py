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
"""
from collections import OrderedDict

import torch

from . import kernels as K
from .nets import FNET_CPAD, GEN_CPAD, FNet, Generator
from .params import ParamStore, fnet_spec, generator_spec, init_values


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
        self.pre_inputs = torch.zeros(batch, h, w, 3, device=self.dev)
        self.pre_gen = torch.zeros(batch, 4 * h, 4 * w, 3, device=self.dev)
        self.use_graph, self.graph = use_graph, None

    def load(self, values):
        """values: TF-variable-name -> tensor for the 'generator' and 'fnet' scopes (main.py:221-224)."""
        self.ps.load(values)

    def reset(self):
        self.pre_inputs.zero_()
        self.pre_gen.zero_()

    def _program(self):
        B, h, w = self.B, self.h, self.w
        fin = K.concat2_pad(self.pre_inputs, self.frame,
                            torch.empty(B, h, w, FNET_CPAD, device=self.dev, dtype=self.act_dtype))
        flow, _ = self.Fn.forward(fin, keep=False)                                # [B, h-h%8, w-w%8, 2]
        x_in = torch.empty(B, h, w, GEN_CPAD, device=self.dev, dtype=self.act_dtype)
        K.warp_s2d_forward(self.pre_gen, flow, self.frame, x_in, 1.0, 0.0)        # state already in [0,1]
        # generator; the fused bicubic / preprocess epilogue writes deprocess(frame) straight into the recurrent state
        # (the warp kernel above has consumed the old state by then: stream order)
        self.G.forward(x_in, keep=False, out=False, state=self.pre_gen)
        self.pre_inputs.copy_(self.frame)

    def step(self, frame=None):
        """frame: [B,h,w,3] fp32 in [0,1] (device tensor).  Returns the HR frame [B,4h,4w,3] in [0,1]
        (a view of the recurrent state: copy it if you keep it across steps)."""
        if frame is not None:
            self.frame.copy_(frame, non_blocking=True)
        if not self.use_graph:
            self._program()
        else:
            if self.graph is None:
                keep = (self.pre_inputs.clone(), self.pre_gen.clone())
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self._program()
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                self.pre_inputs.copy_(keep[0])
                self.pre_gen.copy_(keep[1])
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self._program()
            self.graph.replay()
        return self.pre_gen
