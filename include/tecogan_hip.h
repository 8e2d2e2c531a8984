/*
 * tecogan_hip.h -- C ABI of libtecogan_hip.so (MI355X / gfx950 only).
 *
 * The reference (thunil/TecoGAN) has no native ABI: its operator surface is the
 * Python layer lib/ops.py + the TF1 ops it calls.  Each entry point below names
 * the reference interface it replaces (file:line in /root/reference).
 *
 * Conventions (all entry points):
 *   - return 0 (TG_OK) or a negative TG_E* code; never throw, never allocate,
 *     never synchronise; enqueue-only on `stream` (a hipStream_t passed as
 *     void*), hence hipGraph-capturable;
 *   - every pointer is a caller-owned DEVICE pointer, NHWC-contiguous, 16-byte
 *     aligned; dtype codes: TG_F32 = 0, TG_BF16 = 1;
 *   - re-entrant: the only process-wide state is (a) the thread-local last-error string, (b) one-time, call_once-guarded
 *     kernel attributes, (c) A/B environment switches (TG_*) read once on first use and constant afterwards, and
 *     (d) the opt-in launch profiler's record list (tg_prof_*, mutex-protected).
 */
#ifndef TECOGAN_HIP_H
#define TECOGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_OK        0
#define TG_EINVAL   -1   /* bad argument / unsupported shape */
#define TG_ELAUNCH  -2   /* hipLaunch error (see tg_last_error_string) */

#define TG_F32   0
#define TG_BF16  1

/* activation codes for conv epilogues / pointwise */
#define TG_ACT_NONE    0
#define TG_ACT_RELU    1   /* tf.nn.relu                      lib/frvsr.py:53,63,74,77 */
#define TG_ACT_LRELU   2   /* lrelu(x, alpha)                 lib/ops.py:84-85        */
#define TG_ACT_TANH    3   /* tanh(x) * alpha                 lib/frvsr.py:39         */
#define TG_ACT_SIGMOID 4   /* tf.nn.sigmoid                   lib/Teco.py:72          */

int         tg_version(void);
const char* tg_last_error_string(void);

/* Built-in launch profiler (measurement mode, off by default; no counterpart in the reference, which only prints
 * wall-clock rates: main.py:270,407-411).  While enabled, every convolution / weight-gradient / warp launch of this
 * library carries a start/stop event pair (the dispatch's own timestamps, as rocprofv3 --kernel-trace reads them);
 * tg_prof_collect synchronises, aggregates per kernel name and clears the record list.  Eager streams only. */
typedef struct tg_prof_entry {
  char    name[96];      /* kernel template + tile, e.g. "conv3x3_tile<bf16,bf16,16,64>" */
  int64_t calls;
  double  total_us;      /* sum of dispatch durations */
  double  flops;         /* sum of algorithmic FLOPs (2*M*N*K), 0 for data-movement kernels */
  double  bytes;         /* sum of algorithmic bytes (inputs + outputs + weights once) */
} tg_prof_entry;
int tg_prof_enable(int on);
int tg_prof_collect(tg_prof_entry* out, int max_entries, int* count /* total distinct kernels */);
/* *dst (device, uint64) = the device's 100 MHz wall clock when the stream reaches this point; capturable, so it can mark
 * segment boundaries inside a replayed hipGraph. */
int tg_prof_stamp(void* dst, void* stream);
/* Census of a captured hipGraph (a hipGraph_t passed as void*): number of nodes and of kernel nodes.  Host-only, no launch.  The
 * data-parallel engine (no counterpart in the reference, which has no multi-GPU code) checks with it that its captured exchange
 * segments hold the collective's kernels: RCCL elides them for a one-rank communicator and the captured graph is empty. */
int tg_graph_node_count(void* graph, int* total, int* kernels);

/* ------------------------------------------------------------------------ *
 * Convolution engine (implicit GEMM on MFMA).
 * Replaces slim.conv2d / slim.conv2d_transpose as called by
 *   conv2       lib/ops.py:47-56   (k3 s1, k4 s2; SAME)
 *   conv2_tran  lib/ops.py:35-44   (k3 s2 SAME)
 *   denselayer  lib/ops.py:96-103  (as a 1x1 conv)
 * and their input-gradients.
 *
 * mode 0 (gather):      out[n,oy,ox,:] = sum_{kh,kw} in[n, oy*s-pad_t+kh, ox*s-pad_l+kw, :] . W[kh,kw]
 * mode 1 (transposed):  out[n,oy,ox,:] = sum_{kh,kw : (oy+pad_t-kh)%s==0 ...}
 *                                         in[n,(oy+pad_t-kh)/s,(ox+pad_l-kw)/s,:] . W[kh,kw]
 *   conv fwd            = gather     with W^T copy  (weights_t)
 *   conv bwd_data       = transposed with the HWIO weights as stored by TF
 *   conv_transpose fwd  = transposed with the [kh,kw,Cout,Cin] weights as stored by TF, pad 0
 *   conv_transpose bwd  = gather     with W^T copy, pad 0
 * The weight operand is always laid out [KH*KW][Cout][Cin] with Cin (the
 * reduction channel) contiguous; tg_pack_weights produces the transposed copy.
 *
 * Epilogue, in this order:  v = acc + bias[co];  v = act(v);  v += res;
 *                           v *= act'(aux)   (mask_act: RELU -> aux>0, LRELU -> aux>0 ? 1 : mask_alpha)
 * ------------------------------------------------------------------------ */
typedef struct tg_conv_desc {
  int32_t N, Hin, Win, Cin;        /* the tensor that is read                      */
  int32_t Hout, Wout, Cout;        /* the tensor that is written                   */
  int32_t KH, KW, stride, pad_t, pad_l;
  int32_t mode;                    /* 0 gather, 1 transposed                        */
  int32_t in_dtype, out_dtype;     /* weights share in_dtype; res/aux share out_dtype */
  int32_t act;  float act_alpha;
  int32_t mask_act; float mask_alpha;
  int32_t flags;                   /* TG_CONV_* scheduling hints; 0 = default (results never depend on them) */
} tg_conv_desc;

/* The launch will run beside the latency-bound recurrent chain on another stream: pick tile shapes whose LDS /
 * register footprint leaves room for a co-resident chain workgroup (3x3: <8,64> = 109 KB / ~300 registers instead of
 * <16,64> = 130 KB / ~400; measured: 87 % of the chain hides under VGG-sized convs, profiles/r02a_overlap.txt). */
#define TG_CONV_COEXIST 1

int tg_conv_forward(const tg_conv_desc* d, const void* in, const void* weight /*[KH*KW][Cout][Cin]*/,
                    const float* bias /*nullable*/, const void* res /*nullable*/,
                    const void* aux /*nullable*/, void* out, void* stream);

/* The throughput-regime 3x3 64 -> 64 bf16 convolution with the weight operand in FRAGMENT order (tg_pack_weights_frag, dst_t for
 * the forward conv): out = act(conv3x3(x, W) + b) [+ res] -- the generator's residual-block convs at inference resolution
 * (reference lib/frvsr.py:50-57 through main.py:204; 32 launches per 1080p frame).  Same kernel and arithmetic as
 * tg_conv_forward takes for this shape (csrc/conv3x3_ws.hip), bit-identical results; the weight prologue reads whole cache lines.
 * TG_EINVAL below 256 tiles of 8x16 pixels (latency regime: tg_resblock / tg_conv_forward).  act: TG_ACT_NONE/RELU/LRELU. */
int tg_conv3x3_c64_frag(const void* x, const void* w_frag, const float* bias /*nullable*/, const void* res /*nullable*/, void* out,
                        int N, int H, int W, int act, float act_alpha, void* stream);

/* Wide 3x3 stride-1 SAME bf16 convolutions of the FROZEN perceptual-loss network with the weight operand in fragment order
 * (csrc/conv3x3_wr.hip): VGG-19 conv2_2 ... conv4_4 of reference lib/ops.py:319-327 as called through lib/Teco.py:5-24,174-178,
 * and their input gradients under tf.gradients (lib/Teco.py:441-449).  Same arithmetic, epilogue and results (bit-identical) as
 * tg_conv_forward for the descriptor `d` (mode 0 with the W^T copy, mode 1 with the natural copy); the weight fragments stream
 * global -> registers, only the activation halo goes through LDS.  Cin % 32 == 0, Cin >= 64, Cout % 64 == 0.
 *   tg_pack_wide_frag: w [9][Cout][Cin] bf16 (the operand tg_conv_forward takes) -> w_frag[Cout/16][Cin/32][9][64][8] with
 *     w_frag[g][c][t][l][j] = w[flip ? 8 - t : t][16 g + l % 16][32 c + 8 (l / 16) + j]; flip = 1 for the input-gradient
 *     operand (d->mode == 1: the taps are mirrored in the copy, the kernel sees one direction only).  Once per weight load:
 *     the network is frozen (lib/Teco.py:421 collects generator / fnet / discriminator variables only).
 * Images of exactly 8 x 8 pixels (VGG conv5_x) run as PACKED tiles -- two whole images per 8 x 16 tile, each with its own zero
 * border -- and, when the launch has fewer than one wave per SIMD, with the input channels SPLIT over ksplit groups of waves
 * whose partial sums meet in LDS in a fixed order (deterministic; not bit-identical to tg_conv_forward's single sum).
 *   tile_rows: 0 = chosen from the launch size, 8 or 16 = forced; ksplit: 0 = chosen from the launch size, 1 / 2 / 4 = forced
 *   (8-row tiles only; must divide Cin / 32). */
int tg_pack_wide_frag(const void* w, void* w_frag, int Cout, int Cin, int flip, void* stream);
int tg_conv3x3_wide_frag(const tg_conv_desc* d, const void* in, const void* w_frag, const float* bias /*nullable*/,
                         const void* res /*nullable*/, const void* aux /*nullable*/, void* out, int tile_rows, int ksplit,
                         void* stream);

/* discriminator_F's four `conv2(net, 4, C, 2)` layers (reference lib/Teco.py:52-66; conv2 = lib/ops.py:47-56, slim.conv2d k4 s2
 * SAME) and their input gradients under tf.gradients (lib/Teco.py:393-449), bf16 (csrc/conv4x4s2.hip): the same result as
 * tg_conv_forward for the descriptor `d` up to the summation order (same products, fp32 accumulation).
 *   d->mode 0: out [N,H/2,W/2,Cout] = conv(in [N,H,W,Cin]) (+ bias, none / ReLU / LeakyReLU, + res); H, W even, pad 1
 *   d->mode 1: out [N,2H,2W,Cout]   = input gradient from dY = in [N,H,W,Cin] (+ res, * act'(aux)); the four output phases run as
 *              2x2-tap stride-1 convolutions over one halo of dY
 * w_frag = tg_pack_taps_frag(w, 16, Cout, Cin) of the [16][Cout][Cin] operand tg_conv_forward takes for the same descriptor
 * (mode 0: the W^T copy, mode 1: the HWIO weights as stored).  Cin % 32 == 0, Cout % 64 == 0.
 *   tg_pack_taps_frag: w [taps][Cout][Cin] bf16 -> w_frag[Cout/16][Cin/32][taps][64][8],
 *     w_frag[g][c][t][l][j] = w[t][16 g + l % 16][32 c + 8 (l / 16) + j]  (a wave's weight load = 1 KiB contiguous).
 *   bn_stats (mode 0, nullable; needs act none and no residual): [TG_BN_STAT_REPLICAS][2][Cout] fp32, ZEROED by the caller; the
 *     launch adds the per-channel mean and second moment (E[v], E[v^2]) of v = conv + bias over all N Ho Wo positions, from the
 *     fp32 accumulators, spread over the replicas (one address per channel would serialise 768 workgroups' atomics) -- the batch
 *     statistics slim.batch_norm takes next (lib/ops.py:88-90); tg_bn_lrelu_forward(prezeroed = 2) sums the replicas into the
 *     first one as [mean, biased variance] and skips its own two reduction launches. */
#define TG_BN_STAT_REPLICAS 16
int tg_pack_taps_frag(const void* w, void* w_frag, int taps, int Cout, int Cin, void* stream);
/* The same for `count` tensors in one launch (the per-step refresh of the discriminator's eight operands): tab (device) = 5 x int64
 * per tensor {source element offset -- in src_t, or in src_n when bit 62 is set --, destination element offset in dst, taps, Cout,
 * Cin}; all offsets multiples of 8 elements. */
int tg_pack_taps_frag_multi(const void* src_t, const void* src_n, void* dst, const int64_t* tab, int count, void* stream);
int tg_conv4x4s2_frag(const tg_conv_desc* d, const void* in, const void* w_frag, const float* bias /*nullable*/,
                      const void* res /*nullable*/, const void* aux /*nullable*/, void* out, float* bn_stats /*nullable*/,
                      void* stream);

/* One residual block of generator_F, out = x + conv3x3(relu(conv3x3(x, W1) + b1), W2) + b2 (reference lib/frvsr.py:50-57), as ONE
 * launch in the THROUGHPUT regime (csrc/resblock_thr.hip): the inference step's 16 blocks at [1,270,480,64] (main.py:195-216).
 * bf16, 64 channels; w1_frag / w2_frag = the fragment-order forward operands written by tg_pack_weights_frag (dst_t), the ones
 * tg_conv3x3_c64_frag takes.  The intermediate never leaves the CU (rounded to bf16 once, as the two-launch path stores it); the
 * result equals two tg_conv3x3_c64_frag launches up to the summation order inside each conv.  x and out must not alias. */
int tg_resblock_c64_thr(const void* x, const void* w1_frag, const float* b1 /*nullable*/, const void* w2_frag,
                        const float* b2 /*nullable*/, void* out, int N, int H, int W, void* stream);

/* Weight gradient of the gather-form convolution described by `d`
 * (X = the tensor that is gathered, [N,Hin,Win,Cin]; Y = per-output-pixel tensor
 * [N,Hout,Wout,Cout]):   dW[tap][cx][cy] += sum_m X[m@tap][cx] * Y[m][cy]   (fp32 atomics)
 *                        dbias[cy]      += sum_m Y[m][cy]                  (if dbias != NULL)
 * conv:            X = layer input, Y = dOut  -> dW is HWIO            (lib/ops.py:47-56)
 * conv_transpose:  X = dOut,        Y = layer input -> dW is [kh,kw,Cout,Cin] (lib/ops.py:35-44),
 *                  dbias must then be reduced over X (use tg_colsum).
 * x_dtype/y_dtype: TG_F32 / TG_BF16; accumulation and dW are always fp32. */
int tg_conv_wgrad(const tg_conv_desc* d, const void* x, int x_dtype, int ldx /*channel stride of x, 0 = Cin*/,
                  const void* y, int y_dtype, int ldy /*0 = Cout*/, float* dw, float* dbias /*nullable*/,
                  void* stream);

/* The same for `groups` layers of identical geometry in one call (the 2*num_resblock 64->64 convs of generator_F,
 * lib/frvsr.py:50-57, whose weight gradients are all due at the end of the BPTT chain): x[g], y[g], dw[g], dbias[g]
 * (dbias nullable as a whole or per entry) are HOST arrays of device pointers.  Up to 40 bf16 stride-1 3-wide layers
 * become ONE launch (per-launch fixed cost paid once, lower split-K degree per layer); anything else runs as
 * `groups` ordinary tg_conv_wgrad launches -- the result is the same either way. */
int tg_conv_wgrad_grouped(const tg_conv_desc* d, int groups, const void* const* x, int x_dtype, int ldx,
                          const void* const* y, int y_dtype, int ldy, float* const* dw, float* const* dbias,
                          void* stream);
/* The same plus ONE more layer of the same spatial geometry and output width with FEWER input channels -- generator_F's input conv
 * (51 channels in a 56-channel pixel, lib/frvsr.py:47-49) beside the 2 x num_resblock trunk convs: X pixels of ldx_extra channels,
 * dW with cin_extra rows per tap.  One launch where the transpose-read kernel applies (bf16, 64 -> 64, W % 32 == 0, H % 8 == 0);
 * otherwise tg_conv_wgrad_grouped + tg_conv_wgrad. */
int tg_conv_wgrad_grouped_plus(const tg_conv_desc* d, int groups, const void* const* x, int x_dtype, int ldx, const void* const* y,
                               int y_dtype, int ldy, float* const* dw, float* const* dbias /*nullable*/, const void* x_extra,
                               int ldx_extra, int cin_extra, const void* y_extra, float* dw_extra, float* dbias_extra /*nullable*/,
                               void* stream);

/* Weight gradients of `groups` layers of DIFFERENT geometry in one call (descs[g], ldx[g], ldy[g] per layer; ld* = 0 means the
 * channel count): the 14 convolutions of fnet, reference lib/frvsr.py:4-41 under tf.gradients (lib/Teco.py:441-449).  bf16 3x3
 * stride-1 SAME layers leave as ONE launch; any other mix falls back to one tg_conv_wgrad per layer.  Same accumulation
 * semantics as tg_conv_wgrad. */
int tg_conv_wgrad_multi(const tg_conv_desc* descs, int groups, const void* const* x, int x_dtype, const int* ldx,
                        const void* const* y, int y_dtype, const int* ldy, float* const* dw, float* const* dbias /*nullable*/,
                        void* stream);

/* out[c] += sum over rows of x[rows][C]  (bias gradient helper) */
int tg_colsum(const void* x, int dtype, int64_t rows, int C, float* out, void* stream);

/* Weight re-layout for `count` tensors described by the device table `tab` (7 x int64 per tensor:
 * src offset, dst offset (elements), taps, A, B, Apad, Bpad).  src is [tap][A][B] fp32;
 * transpose=1: dst[tap][b][a_pad] ; transpose=0: dst[tap][a_pad][b_pad] ; rows a >= A / columns b >= B are zero
 * (channel padding of the first layers: 51->56, 6->8, 27->32, 3->8). */
int tg_pack_weights(const float* src_base, void* dst_base, int dst_dtype, const int64_t* tab,
                    int count, int transpose, void* stream);
/* The same for BOTH layouts in one launch (dst_t = [tap][out][in], dst_n = [tap][in][out]): the per-step refresh of the MFMA
 * weight copies after the three Adam updates. */
int tg_pack_weights_both(const float* src_base, void* dst_t, void* dst_n, int dst_dtype, const int64_t* tab, int count, void* stream);

/* ------------------------------------------------------------------------ *
 * Fused recurrent input builder:
 *   tf.contrib.image.dense_image_warp(pre_HR, upscale_four(4*flow_lr))    lib/Teco.py:113,140 main.py:212-215
 *   -> deprocess (scale,shift)                                            lib/Teco.py:143
 *   -> space-to-depth(4)                                                  lib/Teco.py:145-148 main.py:201
 *   -> concat(LR frame, .)                                                lib/Teco.py:150     main.py:202
 * out[b,i,j,0:3] = lr ; out[b,i,j,3+(dy*4+dx)*3+c] = warp(pre)[b,4i+dy,4j+dx,c]*scale+shift ;
 * out[...,51:Cpad] = 0.   pre == NULL writes zeros (first frame, lib/Teco.py:127-129).
 * flow_lr is [B,hf,wf,2] with hf<=h, wf<=w: rows/cols beyond are SYMMETRIC-mirrored (main.py:188-190,212).
 * ------------------------------------------------------------------------ */
int tg_warp_s2d_forward(const float* pre /*[B,4h,4w,3] nullable*/, const float* flow_lr /*nullable iff pre NULL*/,
                        const float* lr /*[B,h,w,3]*/, void* out /*[B,h,w,Cpad]*/, int out_dtype,
                        int B, int h, int w, int hf, int wf, int Cpad, float scale, float shift,
                        float* warped /*[B,4h,4w,3] nullable: also store the warped frame*/, void* stream);

/* Backward of the above.  d_pre += scatter (atomics; caller zero-fills or pre-loads),
 * d_flow_lr += flow gradient (through alpha and upscale_four(4*.)), hf==h && wf==w required. */
int tg_warp_s2d_backward(const void* d_out /*[B,h,w,Cpad]*/, int dtype, const float* pre, const float* flow_lr,
                         float* d_pre, float* d_flow_lr /*nullable*/, int B, int h, int w, int Cpad,
                         float scale, void* stream);

/* Plain dense_image_warp (lib/Teco.py:120,224,254): out = warp(img[B,H,W,C], flow[B,H,W,2]). */
int tg_warp_forward(const float* img, const float* flow, float* out, int B, int H, int W, int C, void* stream);
int tg_warp_backward(const float* d_out, const float* img, const float* flow, float* d_img /*+= atomics, nullable*/,
                     float* d_flow /*=, nullable*/, int B, int H, int W, int C, void* stream);

/* upscale_four (lib/ops.py:126-163) == legacy bilinear x4 (lib/Teco.py:244); out = gain*up4(in). */
int tg_upscale4_forward(const float* in, float* out, int B, int h, int w, int C, float gain, void* stream);
int tg_upscale4_backward(const float* d_out, float* d_in, int B, int h, int w, int C, float gain, void* stream);

/* slim.max_pool2d 2x2 s2 VALID (lib/ops.py:92-93).  bwd routes to the first max in scan order. */
int tg_maxpool2_forward(const void* in, void* out, int dtype, int N, int H, int W, int C, void* stream);
/* act/alpha: derivative of the activation that produced `in`, fused into the routed gradient.
 * add (nullable, [N,H,W,C]): a second gradient w.r.t. `in` itself (a VGG feature tap, lib/Teco.py:346-352):
 * d_in = (route(d_out) + add) * act'(in). */
int tg_maxpool2_backward(const void* in, const void* d_out, void* d_in, int dtype, int N, int H, int W, int C,
                         int act, float alpha, const void* add, void* stream);

/* tf.image.resize_images x2, legacy bilinear (lib/frvsr.py:21-22). */
int tg_upsample2_forward(const void* in, void* out, int dtype, int N, int H, int W, int C, void* stream);
int tg_upsample2_backward(const void* d_out, void* d_in, int dtype, int N, int H, int W, int C,
                          const void* y /*nullable: activation output to mask with*/, int act, float alpha,
                          void* stream);

/* out = (conv_out + bicubic_four(lr)) * 2 - 1   (lib/frvsr.py:81-87, lib/ops.py:166-212).
 * lr is read from the first 3 channels of the generator input buffer [B,h,w,Cpad].
 * state (nullable): also write deprocess(out) = (out + 1) / 2, the recurrent state of the inference loop (main.py:207);
 * out may then be NULL (the inference step keeps only the state). */
int tg_bicubic_add_preprocess(const float* conv_out /*[B,4h,4w,3]*/, const void* gen_in, int in_dtype, int Cpad,
                              float* out /*nullable if state*/, float* state /*nullable*/, int B, int h, int w,
                              void* stream);

/* One residual block of generator_F (reference lib/frvsr.py:50-57: conv3x3 - ReLU - conv3x3 + skip) or the input-gradient
 * chain of the same block (tf.gradients, lib/Teco.py:441-449) as ONE launch -- the latency regime of the training recurrence
 * (csrc/resblock_lat.hip; bf16, C = 64; anything else: TG_EINVAL, run the block as two tg_conv_forward launches).  Every
 * tensor is [N,H,W,64]; w1 / w2 are [9][64][64] = [tap][out][in] of the FIRST / SECOND convolution applied.
 *   mode 0 (forward):        mid = relu(conv(x, w1) + b1)        out = x + conv(mid, w2) + b2
 *                            x = block input, w1 / w2 = the W^T copies of conv_1 / conv_2
 *   mode 1 (input gradient): mid = convT(x, w1) * (aux1 > 0)     out = (x + convT(mid, w2)) [* (aux2 > 0)]
 *                            x = d(block output), w1 / w2 = conv_2's / conv_1's HWIO weights (taps are mirrored inside),
 *                            aux1 = the saved relu(conv_1) output, aux2 (nullable) = the ReLU output that fed the block,
 *                            mid = d(conv_1 pre-activation) -- what conv_1's weight gradient needs; b1 = b2 = NULL
 * mid may be NULL (stateless forward).  Results are bit-identical to the two-launch path.
 * w_frag != 0: w1 / w2 are the FRAGMENT-order copies written by tg_pack_weights_frag (a wave's weight load is one contiguous
 * KiB instead of 16 half cache lines: 6.6 -> 4.4 us per block at [4,32,32,64], profiles/r04b_ab.txt). */
int tg_resblock(int mode, const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                const void* aux1, const void* aux2, void* mid, void* out, int N, int H, int W, int C, int dtype,
                int w_frag, void* stream);
/* The residual TRUNK of generator_F -- `for i in range(1, FLAGS.num_resblock + 1): net = residual_block(net, 64, 1, ...)`,
 * reference lib/frvsr.py:66-70 -- of one frame (mode 0), or the input-gradient chain through the same blocks (mode 1, lib/Teco.py:
 * 441-449), as ONE persistent launch (csrc/resblock_chain.hip): nblocks (1..16) x tg_resblock(w_frag = 1), bit-identical, with the
 * kernel boundary between blocks replaced by a neighbour hand-off of the 2-pixel ring around every 4x4 tile (tagged 8-byte
 * granules, write-through stores, polled with agent-scope loads).  Per-block arrays are HOST arrays of device pointers in
 * PROCESSING order (mode 1: from the last block of the network to the first): w1 / w2 fragment-order weights of the first / second
 * conv applied, b1 / b2 nullable (arrays or entries), aux1 (mode 1 only) the saved relu(conv_1) outputs, mid nullable; aux2_last
 * (nullable) masks the output of the last block processed (mode 1: the ReLU output of the input stage).  x = input of the first
 * block processed; out[k] may not alias it.
 * pre_x (mode 0, nullable): the generator input [N,H,W,pre_cpad] bf16 (51 channels padded to 56, lib/frvsr.py:47-49) -- the
 * input-stage conv + ReLU (lib/frvsr.py:60-63) then runs in the same launch in front of the first block, on the 8x8 region that
 * block needs (no exchange), bit-identical to its own tg_conv_forward launch; pre_w_frag its [tap][64][64] fragment-order copy
 * (input channels zero-padded; tg_pack_weights_frag with Cin in the table), pre_b nullable, pre_out [N,H,W,64] its output
 * (stored: the weight gradients and the input-gradient chain's mask need it); x is ignored then.
 * scratch: the byte count tg_resblock_chain_scratch_bytes reports, device memory, ZEROED ONCE by the caller at allocation and then
 * owned by these calls (control words + granule ring; epochs advance from launch to launch, so it is never cleared again and a
 * captured launch replays); one scratch per stream -- two launches sharing it may not overlap.
 * Needs every workgroup resident: TG_EINVAL when N * ceil(H/4) * ceil(W/4) exceeds the number of compute units (the caller then
 * runs tg_resblock per block).  A workgroup whose wait exceeds ~0.3 s gives up and is counted in ((unsigned*)scratch)[2]
 * (sticky; results are then garbage -- it means a workgroup could not become resident).
 * variant: 0 = default; bit 0: a fifth wave sweeps the ring; bits 1.. = prefetch distance of the weight stream (14 / 28 / 35). */
int tg_resblock_chain_scratch_bytes(int N, int H, int W, int64_t* bytes);
int tg_resblock_chain(int mode, const void* x, int nblocks, const void* const* w1, const float* const* b1,
                      const void* const* w2, const float* const* b2, const void* const* aux1, const void* aux2_last,
                      void* const* mid, void* const* out, void* scratch, const void* pre_x, int pre_cpad, const void* pre_w_frag,
                      const float* pre_b, void* pre_out, int N, int H, int W, int C, int dtype, int variant, void* stream);
/* The residual trunk of generator_F -- reference lib/frvsr.py:50-57,66-70 -- of one frame of the INFERENCE step (main.py:195-216)
 * as ONE persistent launch in the throughput regime (csrc/resblock_plane.hip): nblocks (1..16) x tg_resblock, bit-identical; one
 * workgroup per 16x32-pixel tile keeps the tile's activations in LDS across all blocks, the one-pixel ring around the tile comes
 * from the neighbour workgroups after every conv (tagged granules as in tg_resblock_chain).  w1 / b1 / w2 / b2: HOST arrays of
 * device pointers per block (fragment-order weights, tg_pack_weights_frag; bias arrays or entries nullable); out [N,H,W,64] =
 * the last block's output (may alias x: the input is read before any output is written -- every workgroup stages its tile first).
 * scratch: tg_resblock_plane_scratch_bytes bytes of device memory, ZEROED ONCE at allocation, then owned by these calls (epochs
 * advance from launch to launch; a captured launch replays); one scratch per stream.
 * TG_EINVAL when N * ceil(H/16) * ceil(W/32) exceeds the number of compute units (every workgroup must be resident: run
 * tg_resblock_c64_thr per block then).  Give-ups are counted in ((unsigned*)scratch)[2] (sticky).
 * pre_x (nullable): the generator input [N,H,W,pre_cpad] bf16 (51 channels in 56-channel pixels, lib/frvsr.py:47-49) -- the
 * input-stage conv + ReLU (lib/frvsr.py:60-63) then runs in the same launch in front of the first block: one more conv pass on the
 * tile and one more hand-off; pre_w_frag its [tap][64][64] fragment-order copy (input channels zero-padded: tg_pack_weights_frag with
 * Cin in the table), pre_b nullable; x is ignored then.
 * variant: 0 = default (weight prefetch distance 6 K steps); 1 = 9 steps. */
int tg_resblock_plane_scratch_bytes(int N, int H, int W, int64_t* bytes);
int tg_resblock_plane(const void* x, int nblocks, const void* const* w1, const float* const* b1, const void* const* w2,
                      const float* const* b2, void* out, void* scratch, const void* pre_x, int pre_cpad, const void* pre_w_frag,
                      const float* pre_b, int N, int H, int W, int C, int dtype, int variant, void* stream);
/* Fragment-order bf16 copies of `count` 64 -> 64 3x3 weights (the residual-block convs of lib/frvsr.py:50-57) for tg_resblock:
 * copy[2 tap + kk][wave][lane][j] = W[tap][row = 16 wave + lane % 16][k = 32 kk + 8 (lane / 16) + j]; dst_t: row = output channel
 * (forward operand), dst_n: row = input channel (input-gradient operand).  tab (device): 3 x int64 per tensor -- offset of the
 * HWIO fp32 tensor [3,3,Cin,64] in src_base, offset (elements) of its 36864-element copy in dst_t / dst_n, Cin (<= 64: the
 * generator's input conv has 51; input channels beyond Cin are zero in the copies). */
int tg_pack_weights_frag(const float* src_base, void* dst_t, void* dst_n, const int64_t* tab, int count, void* stream);

/* Input-gradient chain of generator_F's HR tail (reference lib/frvsr.py:73-87 under tf.gradients, lib/Teco.py:441-449) as ONE
 * launch, latency regime of the training recurrence (csrc/hr_bwd_lat.hip; bf16, 64 channels):
 *     g_out = bf16(scale * d_frame), 3 channels zero-padded to 8        [N,2H2,2W2,8]    (what tg_concat2_pad wrote)
 *     g_t2  = bwd_data(output_stage conv 64 -> 3)(g_out) * relu'(t2)   [N,2H2,2W2,64]   (bit-identical to tg_conv_forward's)
 *     g_t1  = bwd_data(conv_tran2, k3 s2)(g_t2) * relu'(t1)            [N,H2,W2,64]
 * d_frame [N,2H2,2W2,3] fp32; w_out = the output conv's HWIO weights with the 3 outputs padded to 8 ([9][64][8], the natural
 * compute copy); w_tr_frag = conv_tran2's [tap][in][out] operand in FRAGMENT order (tg_pack_weights_frag, dst_t). */
int tg_hr_tail_backward(const float* d_frame, float scale, const void* w_out, const void* t2, const void* w_tr_frag,
                        const void* t1, void* g_out, void* g_t2, void* g_t1, int N, int H2, int W2, void* stream);

/* generator_F's transposed convs in the latency regime of the training recurrence (csrc/hr_fwd_lat.hip; bf16, 64 channels), one
 * launch each (reference lib/frvsr.py:73-87):
 *   tg_deconv_lat_forward: y = relu(conv2d_transpose_k3s2(x, W) + b), x [N,H1,W1,64] -> y [N,2H1,2W1,64]
 *   tg_hr_tail_train:      t2 = relu(conv2d_transpose_k3s2(t1, W2) + b2) (stored when t2 != NULL: the backward pass needs it) and
 *                          frame = (conv3x3(t2, W3) + b3 + bicubic_four(LR)) * 2 - 1, frame [N,2H1,2W1,3] fp32;
 *                          state = (frame + 1) / 2 (the inference loop's recurrent state, main.py:207); frame or state may be
 *                          NULL.  With t2 == NULL this is also the inference frame's tail (main.py:195-216).
 * w_frag / w2_frag: the transposed conv's [tap][out][in] operand (TF's [kh,kw,Cout,Cin] as stored) in FRAGMENT order
 * (tg_pack_weights_frag, dst_n); w3 [9][3][64] = the output conv's [tap][out][in] copy; gen_in as in tg_bicubic_add_preprocess. */
int tg_deconv_lat_forward(const void* x, const void* w_frag, const float* bias, void* y, int N, int H1, int W1, void* stream);
/* Input gradient of the same transposed conv in the latency regime (csrc/hr_bwd_lat.hip): dx = bwd_data(conv2d_transpose k3 s2)(dy)
 * [* relu'(aux)], dy [N,2H,2W,64] bf16 -> dx [N,H,W,64] bf16; w_frag = the [tap][in][out] operand in fragment order
 * (tg_pack_weights_frag, dst_t); aux nullable. */
int tg_deconv_lat_backward(const void* dy, const void* w_frag, const void* aux, void* dx, int N, int H, int W, void* stream);
int tg_hr_tail_train(const void* t1, const void* w2_frag, const float* b2, const void* w3, const float* b3, const void* gen_in,
                     int Cpad, void* t2 /*nullable*/, float* frame /*nullable*/, float* state /*nullable*/, int N, int H1, int W1,
                     void* stream);

/* Pointwise activation gradient: d_in = scale * d_out * act'(y) with y the activation OUTPUT
 * (TANH: alpha - y*y/alpha ; SIGMOID: y(1-y) ; RELU/LRELU as the conv mask); y == NULL -> scale+cast.
 * d_out and y share in_dtype; d_in has out_dtype. */
int tg_act_backward(const void* d_out, const void* y /*nullable*/, void* d_in, int in_dtype, int out_dtype,
                    int64_t n, int act, float alpha, float scale, void* stream);

/* out[pix][0:Ca] = a ; [Ca:Ca+Cb] = b ; [Ca+Cb:Cpad] = 0  -- tf.concat of lib/Teco.py:108, main.py:209
 * (fnet input) and the channel padding of first-layer inputs. */
int tg_concat2_pad(const float* a, int Ca, const float* b /*nullable*/, int Cb, void* out, int out_dtype, int Cpad,
                   int64_t npix, float scale /*applied to every value*/, void* stream);

/* Frame-major gather of a batch-major sequence: dst[t][b][:] = src[b][idx[t]][:] (src [B][T0][frame_elems],
 * dst [T][B][frame_elems], idx = HOST array of T <= 64 frame indices): the ping-pong extension
 * tf.concat(r, r[:, -2::-1]) of lib/Teco.py:80-85 and the [B,T] -> [T,B] layout of this path in one pass. */
int tg_seq_gather(const float* src, float* dst, int B, int T0, int T, int64_t frame_elems, const int* idx,
                  void* stream);

/* out (=|+=) alpha*a + beta*b   (loss-gradient seeds: lib/Teco.py:320-331) */
int tg_lincomb(const float* a, const float* b /*nullable*/, float* out, int64_t n, float alpha, float beta,
               int accumulate, void* stream);

/* out = x*scale + shift: deprocess (lib/ops.py:19-22) with (0.5, 0.5), preprocess (lib/ops.py:13-16) with (2,-1). */
int tg_affine(const float* x, float* out, int64_t n, float scale, float shift, void* stream);

/* Device-side schedule (lib/Teco.py:95-99,415-417,425,439-440,493-494): advances global_step, the
 * lr decay, each optimiser's Adam bias correction and the EMA(0.99)/tf.cond D-gate; fills
 * hyper[k] = {lr_t, beta1, beta2, eps, gate, lr, 0, 0} for tg_adam_tf.  state layout: schedule.hip. */
int tg_schedule_step(double* state, float* hyper, int nopt, int gated_opt, const float* t_balance /*nullable*/,
                     float beta1, float beta2, float eps, void* stream);
/* Fade-in factor of the adversarial / layer losses, reference lib/Teco.py:379-380:
 *   out[0] = min(rmax, r0 + add * state[0])   with state[0] = the device-side global step of tg_schedule_step,
 * so that a captured step replays with a changing factor (consumed by tg_gan_losses / tg_l1_loss through their *_dev scalars). */
int tg_dt_ratio(const double* state, float r0, float add, float rmax, float* out, void* stream);

/* slim.batch_norm(train, scale=False, eps) + LeakyReLU (lib/ops.py:88-90, lib/Teco.py:38-39).
 * stats: [2][C] fp32 (mean, biased var) written by forward, read by backward. */
int tg_bn_lrelu_forward(const void* x, void* y, int dtype, int64_t rows, int C, const float* beta, float eps,
                        float alpha, float* stats, float* moving /*[2][C] nullable, decay .9*/,
                        int prezeroed /*1: caller guarantees stats == 0 on entry (no memset node); 2 (bf16, C % 8 == 0): stats
                        = [TG_BN_STAT_REPLICAS][2][C] holding partial [mean, second moment] of x (tg_conv4x4s2_frag's bn_stats):
                        one tiny launch sums them into stats[0] = [mean, biased variance], the two reduction launches are
                        skipped*/, void* stream);
int tg_bn_lrelu_backward(const void* x, const void* y, const void* d_y, void* d_x, int dtype, int64_t rows, int C,
                         const float* stats, float eps, float alpha, float* d_beta /*+=*/,
                         float* ws /*[2][C] scratch*/, int prezeroed /*1: ws == 0 on entry*/, void* stream);

/* tf.train.AdamOptimizer step over a flat fp32 buffer (lib/Teco.py:425,439-440), TF flavour:
 * p -= lr_t * m / (sqrt(v)+eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t) read from `hyper` = device
 * {lr_t, beta1, beta2, eps, gate}; gate==0 skips the update (the tf.cond D-gate, lib/Teco.py:493-494). */
int tg_adam_tf(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float grad_scale,
               void* stream);

/* Loss reductions (lib/Teco.py:296,322,331,347-352,367): out[0] += scale * sum(...) */
int tg_sum_sq_diff(const void* a, const void* b, int dtype, int64_t n, float scale, float* out, void* stream);
int tg_sum_abs_diff(const void* a, const void* b, int dtype, int64_t n, float scale, float* out, void* stream);

/* ------------------------------------------------------------------------ *
 * TecoGAN losses (forward value + gradient seed in one pass) and the fused discriminator input.
 * ------------------------------------------------------------------------ */
/* Ping-pong L1 (lib/Teco.py:362-370) on the frame-major sequence gen[T][frame_elems]:
 * pairs (k, T-1-k) for k < npair; loss += loss_scale*sum|a-b|; d_gen[k] += grad_scale*sign, d_gen[T-1-k] -= . */
int tg_pingpong(const float* gen, float* d_gen, int T, int npair, int64_t frame_elems, float loss_scale,
                float grad_scale, float* loss, void* stream);

/* VGG input transform (lib/Teco.py:9-10): ((x+1)/2)*255 - VGG_MEAN, zero-padded to Cpad channels; and its
 * gradient d_x += 127.5 * d_out[..., :3]. */
int tg_vgg_preprocess_forward(const float* x, void* out, int out_dtype, int64_t npix, int Cpad, void* stream);
int tg_vgg_preprocess_backward(const void* d_out, int dtype, float* d_x, int64_t npix, int Cpad, void* stream);

/* Cosine feature loss of one VGG tap (lib/Teco.py:15-23,346-352): cos_sum += cos_scale * sum_pix cos(g,t),
 * d_g = grad_scale * d cos / d g (nullable). */
int tg_cosine_loss(const void* g, const void* t, int dtype, int64_t npix, int C, float cos_scale, float grad_scale,
                   float* cos_sum, void* d_g /*nullable*/, void* stream);

/* L1 layer loss (lib/Teco.py:291-302): loss += loss_scale*sum|r-f|; d_f = -grad_scale*sign(r-f) (nullable). */
int tg_l1_loss(const void* r, const void* f, int dtype, int64_t n, float loss_scale, float grad_scale,
               const float* grad_scale_dev /*nullable: device scalar multiplied into grad_scale (dt_ratio)*/, float* loss,
               void* d_f /*nullable*/, void* stream);

/* Adversarial losses (lib/Teco.py:374-399) on the sigmoid outputs of both D passes:
 * out = {t_adversarial_loss, t_discrim_loss, t_balance, mean(real), mean(fake)};
 * d_real_D/d_fake_D = gradient of t_discrim_loss, d_fake_G = gradient of adv_weight * t_adversarial_loss. */
int tg_gan_losses(const float* real, const float* fake, int n, float eps, float adv_weight,
                  const float* adv_weight_scale_dev /*nullable: device scalar multiplied into adv_weight (dt_ratio)*/,
                  float* out, float* d_real_D, float* d_fake_D, float* d_fake_G, void* stream);

/* Fused discriminator input (lib/Teco.py:180-272): for triplet k of sample b (frames 3k..3k+2 of the frame-major
 * sequence frames[T][B][4h][4w][3]): before-warp | warp with {up4(4*flow_pre[idx_pre[k]]), 0, up4(4*flow_nxt[idx_nxt[k]])}
 * centre-cropped by `off` (zero border) | legacy-bilinear x4 LR context, channel = c*3+t, zero-padded to Cpad.
 * merge=0 (Dt only, lib/Teco.py:249-250): only the warped block, output cropped to (4h-2off)^2.
 * idx_pre / idx_nxt are HOST int arrays of length nt (<=16).  tb index = k*B + b. */
int tg_pack_d_input_forward(const float* frames, const float* lr, const float* flow_pre, const float* flow_nxt,
                            const int* idx_pre, const int* idx_nxt, void* out, int out_dtype, int B, int h, int w,
                            int nt, int off, int merge, int Cpad, void* stream);
/* d_frames += gradient w.r.t. the frames (fp32 atomics); no gradient to the flows (stop_gradient, lib/Teco.py:214). */
int tg_pack_d_input_backward(const void* d_out, int dtype, const float* frames, const float* flow_pre,
                             const float* flow_nxt, const int* idx_pre, const int* idx_nxt, float* d_frames, int B,
                             int h, int w, int nt, int off, int merge, int Cpad, void* stream);

/* ------------------------------------------------------------------------ *
 * Data step before the path / output step after it.
 * ------------------------------------------------------------------------ */
/* tf_data_gaussDownby4 (lib/ops.py:347-367): depthwise k x k Gaussian (HOST array `weights`, k*k, row-major, k <= 11),
 * stride 4, VALID: hr [N,H,W,3] fp32 -> lr [N,(H-k)/4+1,(W-k)/4+1,3]; target (nullable) additionally receives
 * preprocess(hr[:, border:border+4h, border:border+4w]) = 2x-1 (lib/dataloader.py:306-332) in the same pass. */
int tg_gauss_down4_preprocess(const float* hr, float* lr, float* target /*nullable*/, int N, int H, int W, int k,
                              const float* weights /*host*/, int border, void* stream);

/* save_img (lib/ops.py:521-523): out = uint8(clip(frame * 255, 0, 255)) (truncation), RGB or BGR byte order. */
int tg_frame_to_u8(const float* frame /*[npix][3] in [0,1]*/, unsigned char* out /*[npix][3]*/, int64_t npix, int bgr,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TECOGAN_HIP_H */
