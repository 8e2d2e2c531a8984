// TecoGAN loss kernels and the fused discriminator-input builder (gfx950; all HBM/latency bound).
// Every loss kernel is forward + gradient-seed in one pass: the training step needs both, and the
// tensors are read exactly once.  Reference formulas: lib/Teco.py:180-272 (D inputs), :275-313 (layer
// loss), :339-359 (VGG cosine loss), :362-372 (ping-pong), :374-417 (adversarial / balance).
#include "common.h"

__device__ __forceinline__ void block_atomic_add(float v, float* dst) {
  __shared__ float red_[4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red_[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(dst, red_[0] + red_[1] + red_[2] + red_[3]);
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Ping-pong loss (lib/Teco.py:362-370): gen is frame-major [T][frame]; pairs (k, T-1-k), k < npair.
//   loss += loss_scale * sum |a-b| ;  d_gen[k] += g*sign(a-b) ; d_gen[T-1-k] -= g*sign(a-b)
__global__ __launch_bounds__(256) void pingpong_kernel(const float* __restrict__ gen, float* __restrict__ d_gen, int T,
                                                       int npair, int64_t frame, float loss_scale, float grad_scale,
                                                       float* __restrict__ loss) {
  const int64_t n = (int64_t)npair * frame;
  float s = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = e / frame, i = e % frame;
    const int64_t ia = k * frame + i, ib = (int64_t)(T - 1 - k) * frame + i;
    const float d = gen[ia] - gen[ib];
    s += fabsf(d);
    const float sg = d > 0.f ? grad_scale : (d < 0.f ? -grad_scale : 0.f);
    d_gen[ia] += sg;
    d_gen[ib] -= sg;
  }
  block_atomic_add(s * loss_scale, loss);
}

extern "C" int tg_pingpong(const float* gen, float* d_gen, int T, int npair, int64_t frame_elems, float loss_scale,
                           float grad_scale, float* loss, void* stream) {
  TG_CHECK_ARG(gen && d_gen && loss && T > 1 && npair > 0 && 2 * npair < T + 1 && frame_elems > 0, "bad argument");
  hipLaunchKernelGGL(pingpong_kernel, TG_DET_GRID(grid_1d((int64_t)npair * frame_elems, 256 * 4, 1024)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), gen, d_gen, T, npair, frame_elems, loss_scale, grad_scale, loss);
  TG_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// VGG input (lib/Teco.py:9-10): out[...,c] = ((x+1)/2)*255 - mean[c], zero-padded to Cpad channels.
template <typename TO>
__global__ __launch_bounds__(256) void vgg_pre_fwd_kernel(const float* __restrict__ x, TO* __restrict__ out,
                                                          int64_t npix, int Cpad) {
  const float mean[3] = {123.68f, 116.78f, 103.94f};
  const int64_t n = npix * Cpad;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % Cpad);
    const int64_t pix = e / Cpad;
    Elem<TO>::st(out + e, c < 3 ? ((x[pix * 3 + c] + 1.f) / 2.f) * 255.f - mean[c] : 0.f);
  }
}
// bf16, 8 padded channels (the layout the training step uses): one thread per pixel, one 16-byte store (the element-wise form above
// stores 2 bytes per thread and iteration: 34 us for 48 images of 128 x 128, profiles/r05d_tecogan_bf16_kernel_stats.txt)
__global__ __launch_bounds__(256) void vgg_pre_fwd_x8_kernel(const float* __restrict__ x, uint4* __restrict__ out, int64_t npix) {
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const float r = ((x[pix * 3] + 1.f) / 2.f) * 255.f - 123.68f;
  const float g = ((x[pix * 3 + 1] + 1.f) / 2.f) * 255.f - 116.78f;
  const float b = ((x[pix * 3 + 2] + 1.f) / 2.f) * 255.f - 103.94f;
  out[pix] = make_uint4((uint32_t)f2bf(r) | ((uint32_t)f2bf(g) << 16), (uint32_t)f2bf(b), 0u, 0u);
}
template <typename TI>
__global__ __launch_bounds__(256) void vgg_pre_bwd_kernel(const TI* __restrict__ d_out, float* __restrict__ d_x,
                                                          int64_t npix, int Cpad) {
  const int64_t n = npix * 3;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    d_x[e] += Elem<TI>::ld(d_out + (e / 3) * Cpad + (e % 3)) * 127.5f;
}

extern "C" int tg_vgg_preprocess_forward(const float* x, void* out, int out_dtype, int64_t npix, int Cpad,
                                         void* stream) {
  TG_CHECK_ARG(x && out && npix > 0 && Cpad >= 3, "bad argument");
  dim3 g(grid_1d(npix * Cpad, 256));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (out_dtype == TG_BF16 && Cpad == 8 && ((uintptr_t)out & 15) == 0)
    hipLaunchKernelGGL(vgg_pre_fwd_x8_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, x, (uint4*)out, npix);
  else if (out_dtype == TG_F32) hipLaunchKernelGGL((vgg_pre_fwd_kernel<float>), g, dim3(256), 0, st, x, (float*)out, npix, Cpad);
  else if (out_dtype == TG_BF16) hipLaunchKernelGGL((vgg_pre_fwd_kernel<u16>), g, dim3(256), 0, st, x, (u16*)out, npix, Cpad);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}
extern "C" int tg_vgg_preprocess_backward(const void* d_out, int dtype, float* d_x, int64_t npix, int Cpad,
                                          void* stream) {
  TG_CHECK_ARG(d_out && d_x && npix > 0 && Cpad >= 3, "bad argument");
  dim3 g(grid_1d(npix * 3, 256));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == TG_F32) hipLaunchKernelGGL((vgg_pre_bwd_kernel<float>), g, dim3(256), 0, st, (const float*)d_out, d_x, npix, Cpad);
  else if (dtype == TG_BF16) hipLaunchKernelGGL((vgg_pre_bwd_kernel<u16>), g, dim3(256), 0, st, (const u16*)d_out, d_x, npix, Cpad);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// VGG cosine feature loss (lib/Teco.py:15-23,346-352): per pixel, gh = g/sqrt(sum g^2+1e-12), same for t;
//   cos_sum += cos_scale * sum_pix (gh . th) ;  d_g = grad_scale * (th - gh (gh.th)) / |g|
// One wave per pixel (C = 128..512 channels), shuffle reductions.
template <typename T>
__global__ __launch_bounds__(256) void cosine_loss_kernel(const T* __restrict__ g, const T* __restrict__ t,
                                                          int64_t npix, int C, float cos_scale, float grad_scale,
                                                          float* __restrict__ cos_sum, T* __restrict__ d_g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int64_t pix = (int64_t)blockIdx.x * 4 + wave; pix < npix; pix += (int64_t)gridDim.x * 4) {
    const T* __restrict__ gp = g + pix * C;
    const T* __restrict__ tp = t + pix * C;
    float gg = 0.f, tt = 0.f, gt = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float a = Elem<T>::ld(gp + c), b = Elem<T>::ld(tp + c);
      gg += a * a;
      tt += b * b;
      gt += a * b;
    }
    gg = wave_sum(gg);
    tt = wave_sum(tt);
    gt = wave_sum(gt);
    const float ng = sqrtf(gg + 1e-12f), nt = sqrtf(tt + 1e-12f);
    const float cosv = gt / (ng * nt);
    if (lane == 0) acc += cosv;
    if (d_g) {
      T* __restrict__ dp = d_g + pix * C;
      const float k1 = grad_scale / (ng * nt), k2 = grad_scale * cosv / (ng * ng);
      for (int c = lane; c < C; c += 64) Elem<T>::st(dp + c, k1 * Elem<T>::ld(tp + c) - k2 * Elem<T>::ld(gp + c));
    }
  }
  block_atomic_add(acc * cos_scale, cos_sum);
}

// bf16, C in {64,128,256,512}: LP = C/8 lanes share a pixel (64/LP pixels per wave), every lane owns
// one 16-byte channel octet of g and t, kept in registers for the gradient pass; sub-wave xor reductions.
template <int LP>
__global__ __launch_bounds__(256) void cosine_loss_x8_kernel(const u16* __restrict__ g, const u16* __restrict__ t,
                                                             int64_t npix, float cos_scale, float grad_scale,
                                                             float* __restrict__ cos_sum, u16* __restrict__ d_g) {
  constexpr int PPW = 64 / LP, C = LP * 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LP, oc = lane % LP;
  float acc = 0.f;
  for (int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * PPW; p0 < npix; p0 += (int64_t)gridDim.x * 4 * PPW) {
    const int64_t pix = p0 + sub;
    const bool ok = pix < npix;
    const int64_t off = (ok ? pix : p0) * C + oc * 8;
    float a[8], b[8];
    bf8_unpack(*reinterpret_cast<const uint4*>(g + off), a);
    bf8_unpack(*reinterpret_cast<const uint4*>(t + off), b);
    float gg = 0.f, tt = 0.f, gt = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      gg += a[k] * a[k];
      tt += b[k] * b[k];
      gt += a[k] * b[k];
    }
#pragma unroll
    for (int o = LP / 2; o > 0; o >>= 1) {
      gg += __shfl_xor(gg, o, 64);
      tt += __shfl_xor(tt, o, 64);
      gt += __shfl_xor(gt, o, 64);
    }
    const float ng = sqrtf(gg + 1e-12f), nt = sqrtf(tt + 1e-12f);
    const float cosv = gt / (ng * nt);
    if (ok && oc == 0) acc += cosv;
    if (d_g && ok) {
      const float k1 = grad_scale / (ng * nt), k2 = grad_scale * cosv / (ng * ng);
      float d[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = k1 * b[k] - k2 * a[k];
      *reinterpret_cast<uint4*>(d_g + off) = bf8_pack(d);
    }
  }
  block_atomic_add(acc * cos_scale, cos_sum);
}

extern "C" int tg_cosine_loss(const void* g, const void* t, int dtype, int64_t npix, int C, float cos_scale,
                              float grad_scale, float* cos_sum, void* d_g, void* stream) {
  TG_CHECK_ARG(g && t && cos_sum && npix > 0 && C > 0, "bad argument");
  if (dtype == TG_BF16 && (C == 64 || C == 128 || C == 256 || C == 512) &&
      ((((uintptr_t)g | (uintptr_t)t | (uintptr_t)d_g)) & 15) == 0) {
    hipStream_t sx = static_cast<hipStream_t>(stream);
    const int lp = C / 8;
    // <= 1024 workgroups: every workgroup ends with ONE atomic on the same address and same-address atomics serialise at
    // the memory side (~10 ns each): 8192 of them cost ~100 us, more than the streaming itself
    dim3 gx(grid_1d(npix, 4 * (64 / lp), 1024));
#define TG_COS(LP_) hipLaunchKernelGGL((cosine_loss_x8_kernel<LP_>), TG_DET_GRID(gx), dim3(256), 0, sx, (const u16*)g, (const u16*)t, npix, cos_scale, grad_scale, cos_sum, (u16*)d_g)
    if (lp == 8) TG_COS(8);
    else if (lp == 16) TG_COS(16);
    else if (lp == 32) TG_COS(32);
    else TG_COS(64);
#undef TG_COS
    TG_CHECK_LAUNCH();
  }
  dim3 grid(grid_1d(npix, 4, 1024));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == TG_F32) hipLaunchKernelGGL((cosine_loss_kernel<float>), TG_DET_GRID(grid), dim3(256), 0, st, (const float*)g, (const float*)t, npix, C, cos_scale, grad_scale, cos_sum, (float*)d_g);
  else if (dtype == TG_BF16) hipLaunchKernelGGL((cosine_loss_kernel<u16>), TG_DET_GRID(grid), dim3(256), 0, st, (const u16*)g, (const u16*)t, npix, C, cos_scale, grad_scale, cos_sum, (u16*)d_g);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// L1 feature ("layer") loss (lib/Teco.py:291-302): loss += loss_scale * sum|r-f| ; d_f = -grad_scale*sign(r-f)
template <typename T>
__global__ __launch_bounds__(256) void l1_loss_kernel(const T* __restrict__ r, const T* __restrict__ f, int64_t n,
                                                      float loss_scale, float grad_scale, const float* __restrict__ gscale,
                                                      float* __restrict__ loss, T* __restrict__ d_f) {
  if (gscale) grad_scale *= gscale[0];          // device-side fade-in factor (dt_ratio, lib/Teco.py:379-380,390)
  float s = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float d = Elem<T>::ld(r + e) - Elem<T>::ld(f + e);
    s += fabsf(d);
    if (d_f) Elem<T>::st(d_f + e, d > 0.f ? -grad_scale : (d < 0.f ? grad_scale : 0.f));
  }
  block_atomic_add(s * loss_scale, loss);
}

// bf16, 8 elements (16 bytes) per lane and trip: D's layer maps (lib/Teco.py:291-302; 3.1 M elements at [12,64,64,64]) ran
// 16.8 us on the 2-byte loads of the generic kernel
__global__ __launch_bounds__(256) void l1_loss_x8_kernel(const u16* __restrict__ r, const u16* __restrict__ f, int64_t n8,
                                                         float loss_scale, float grad_scale, const float* __restrict__ gscale,
                                                         float* __restrict__ loss, u16* __restrict__ d_f) {
  if (gscale) grad_scale *= gscale[0];
  float s = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n8; e += (int64_t)gridDim.x * blockDim.x) {
    float a[8], b[8], g[8];
    bf8_unpack(*reinterpret_cast<const uint4*>(r + e * 8), a);
    bf8_unpack(*reinterpret_cast<const uint4*>(f + e * 8), b);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float d = a[k] - b[k];
      s += fabsf(d);
      g[k] = d > 0.f ? -grad_scale : (d < 0.f ? grad_scale : 0.f);
    }
    if (d_f) *reinterpret_cast<uint4*>(d_f + e * 8) = bf8_pack(g);
  }
  block_atomic_add(s * loss_scale, loss);
}

extern "C" int tg_l1_loss(const void* r, const void* f, int dtype, int64_t n, float loss_scale, float grad_scale,
                          const float* grad_scale_dev, float* loss, void* d_f, void* stream) {
  TG_CHECK_ARG(r && f && loss && n > 0, "bad argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == TG_BF16 && n % 8 == 0 && ((((uintptr_t)r | (uintptr_t)f | (uintptr_t)d_f)) & 15) == 0) {
    dim3 g8(grid_1d(n / 8, 256, 1024));
    hipLaunchKernelGGL(l1_loss_x8_kernel, TG_DET_GRID(g8), dim3(256), 0, st, (const u16*)r, (const u16*)f, n / 8, loss_scale, grad_scale,
                       grad_scale_dev, loss, (u16*)d_f);
    TG_CHECK_LAUNCH();
  }
  dim3 grid(grid_1d(n, 256 * 4, 1024));
  if (dtype == TG_F32) hipLaunchKernelGGL((l1_loss_kernel<float>), TG_DET_GRID(grid), dim3(256), 0, st, (const float*)r, (const float*)f, n, loss_scale, grad_scale, grad_scale_dev, loss, (float*)d_f);
  else if (dtype == TG_BF16) hipLaunchKernelGGL((l1_loss_kernel<u16>), TG_DET_GRID(grid), dim3(256), 0, st, (const u16*)r, (const u16*)f, n, loss_scale, grad_scale, grad_scale_dev, loss, (u16*)d_f);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// Adversarial losses (lib/Teco.py:374-399).  out = {t_adversarial, t_discrim, t_balance, mean(real), mean(fake)}
//   t_adv = mean(-log(fake+eps)); t_discrim = mean(-(log(1-fake+eps) + log(real+eps)));
//   t_balance = mean(log(real+eps)) + t_adv.
// Gradient seeds: d_real_D, d_fake_D (of t_discrim), d_fake_G (of adv_weight * t_adv).  One block.
__global__ __launch_bounds__(256) void gan_losses_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                                         int n, float eps, float adv_weight,
                                                         const float* __restrict__ adv_scale, float* __restrict__ out,
                                                         float* __restrict__ d_real_D, float* __restrict__ d_fake_D,
                                                         float* __restrict__ d_fake_G) {
  __shared__ float red[5][4];
  if (adv_scale) adv_weight *= adv_scale[0];    // device-side fade-in factor (dt_ratio, lib/Teco.py:379-384)
  float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const float inv = 1.f / (float)n;
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    const float r = real[e], f = fake[e];
    const float lf = logf(f + eps), l1f = logf(1.f - f + eps), lr = logf(r + eps);
    s[0] += -lf;
    s[1] += -(l1f + lr);
    s[2] += lr;
    s[3] += r;
    s[4] += f;
    d_real_D[e] = -inv / (r + eps);
    d_fake_D[e] = inv / (1.f - f + eps);
    d_fake_G[e] = -adv_weight * inv / (f + eps);
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const float v = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const int k = threadIdx.x;
    red[k][0] = (red[k][0] + red[k][1] + red[k][2] + red[k][3]) * inv;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = red[0][0];
    out[1] = red[1][0];
    out[2] = red[2][0] + red[0][0];
    out[3] = red[3][0];
    out[4] = red[4][0];
  }
}

extern "C" int tg_gan_losses(const float* real, const float* fake, int n, float eps, float adv_weight,
                             const float* adv_weight_scale_dev, float* out, float* d_real_D, float* d_fake_D, float* d_fake_G,
                             void* stream) {
  TG_CHECK_ARG(real && fake && out && d_real_D && d_fake_D && d_fake_G && n > 0, "bad argument");
  hipLaunchKernelGGL(gan_losses_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), real, fake, n, eps,
                     adv_weight, adv_weight_scale_dev, out, d_real_D, d_fake_D, d_fake_G);
  TG_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// Fused discriminator input (lib/Teco.py:180-272).  For triplet (k, b) with frames f = 3k, 3k+1, 3k+2 of the
// frame-major sequence `frames` [T][B][H][W][3]:
//   before[c*3+t] = frames[f_t]                                                    (Teco.py:236-238)
//   warped[c*3+t] = dense_image_warp(frames[f_t], vel_t), vel = {up4(4*flow[pre_k]), 0, up4(4*flow[nxt_k])},
//                   zero outside the centre crop [off, H-off)                      (Teco.py:201-234)
//   hi[c*3+t]     = legacy bilinear x4 of the LR frame f_t                        (Teco.py:240-244)
// merge=1: out[tb][H][W][Cpad] = before | warped | hi | 0 ;  merge=0: out[tb][H-2off][W-2off][Cpad] = warped | 0.
// Thread map as the recurrent warp: 16 lanes per LR cell, flow corners broadcast by shuffles;
// flow_hr is never materialised, stop_gradient on it (Teco.py:214) means no flow gradient here.
struct PackP {
  const float* frames;
  const float* lr;
  const float* flow_pre;
  const float* flow_nxt;
  void* out;
  int B, h, w, nt, off, merge, Cpad;
  int idx_pre[16], idx_nxt[16];
};

struct Tap2 {
  int fy, fx;
  float ay, ax;
};
__device__ __forceinline__ Tap2 tap_of(float qy, float qx, int H, int W) {
  Tap2 t;
  const float fy = fminf(fmaxf(floorf(qy), 0.f), (float)(H - 2)), fx = fminf(fmaxf(floorf(qx), 0.f), (float)(W - 2));
  t.fy = (int)fy;
  t.fx = (int)fx;
  t.ay = fminf(fmaxf(qy - fy, 0.f), 1.f);
  t.ax = fminf(fmaxf(qx - fx, 0.f), 1.f);
  return t;
}
__device__ __forceinline__ float2 up4_flow(const float* __restrict__ flow, int b, int i, int j, int sub, int h, int w) {
  const int i1 = min(i + 1, h - 1), j1 = min(j + 1, w - 1);
  float2 mine = make_float2(0.f, 0.f);
  if (sub < 4)
    mine = *reinterpret_cast<const float2*>(flow + ((int64_t)(b * h + ((sub & 2) ? i1 : i)) * w + ((sub & 1) ? j1 : j)) * 2);
  float2 c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c[k].x = __shfl(mine.x, k, 16);
    c[k].y = __shfl(mine.y, k, 16);
  }
  const float wy = 0.25f * (sub >> 2), wx = 0.25f * (sub & 3);
  const float w0 = (1.f - wy) * (1.f - wx), w1 = (1.f - wy) * wx, w2 = wy * (1.f - wx), w3 = wy * wx;
  return make_float2((c[0].x * 4.f) * w0 + (c[1].x * 4.f) * w1 + (c[2].x * 4.f) * w2 + (c[3].x * 4.f) * w3,
                     (c[0].y * 4.f) * w0 + (c[1].y * 4.f) * w1 + (c[2].y * 4.f) * w2 + (c[3].y * 4.f) * w3);
}

template <typename TO>
__global__ __launch_bounds__(256) void pack_d_fwd_kernel(PackP p) {
  const int H = 4 * p.h, W = 4 * p.w;
  const int64_t ncell = (int64_t)p.nt * p.B * p.h * p.w;
  const int64_t fsz = (int64_t)p.B * H * W * 3, lsz = (int64_t)p.B * p.h * p.w * 3, flsz = (int64_t)p.B * p.h * p.w * 2;
  const int Ho = p.merge ? H : H - 2 * p.off, Wo = p.merge ? W : W - 2 * p.off;
  TO* __restrict__ out = static_cast<TO*>(p.out);
  for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; (gid >> 4) < ncell;
       gid += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cell = gid >> 4;
    const int sub = (int)(gid & 15);
    const int j = (int)(cell % p.w), i = (int)((cell / p.w) % p.h);
    const int b = (int)((cell / ((int64_t)p.w * p.h)) % p.B), k = (int)(cell / ((int64_t)p.w * p.h * p.B));
    const int Y = 4 * i + (sub >> 2), X = 4 * j + (sub & 3);
    const float2 fpre = up4_flow(p.flow_pre + p.idx_pre[k] * flsz, b, i, j, sub, p.h, p.w);
    const float2 fnxt = up4_flow(p.flow_nxt + p.idx_nxt[k] * flsz, b, i, j, sub, p.h, p.w);
    const bool inside = Y >= p.off && Y < H - p.off && X >= p.off && X < W - p.off;
    if (!p.merge && !inside) continue;
    float before[9], warped[9], hi[9];   // statically indexed (registers), channel = c*3 + t
#pragma unroll
    for (int c = 0; c < 9; ++c) before[c] = warped[c] = hi[c] = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const float* __restrict__ fr = p.frames + (int64_t)(3 * k + t) * fsz + (int64_t)b * H * W * 3;
      if (p.merge) {
#pragma unroll
        for (int c = 0; c < 3; ++c) before[c * 3 + t] = fr[((int64_t)Y * W + X) * 3 + c];
        // legacy bilinear x4 of the LR frame (tf.image.resize_images, [TF1] A.4)
        const float* __restrict__ lrp = p.lr + (int64_t)(3 * k + t) * lsz + (int64_t)b * p.h * p.w * 3;
        const int i1 = min(i + 1, p.h - 1), j1 = min(j + 1, p.w - 1);
        const float ya = 0.25f * (sub >> 2), xa = 0.25f * (sub & 3);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float tl = lrp[((int64_t)i * p.w + j) * 3 + c], tr = lrp[((int64_t)i * p.w + j1) * 3 + c];
          const float bl = lrp[((int64_t)i1 * p.w + j) * 3 + c], br = lrp[((int64_t)i1 * p.w + j1) * 3 + c];
          const float top = tl + (tr - tl) * xa, bot = bl + (br - bl) * xa;
          hi[c * 3 + t] = top + (bot - top) * ya;
        }
      }
      if (inside) {
        if (t == 1) {
#pragma unroll
          for (int c = 0; c < 3; ++c) warped[c * 3 + t] = fr[((int64_t)Y * W + X) * 3 + c];
        } else {
          const float2 f = t == 0 ? fpre : fnxt;
          const Tap2 tp = tap_of((float)Y - f.x, (float)X - f.y, H, W);
          const float* __restrict__ p0 = fr + ((int64_t)tp.fy * W + tp.fx) * 3;
          const float* __restrict__ p1 = p0 + (int64_t)W * 3;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float tl = p0[c], tr = p0[3 + c], bl = p1[c], br = p1[3 + c];
            const float top = tp.ax * (tr - tl) + tl, bot = tp.ax * (br - bl) + bl;
            warped[c * 3 + t] = tp.ay * (bot - top) + top;
          }
        }
      }
    }
    const int tb = k * p.B + b;
    const int Yo = p.merge ? Y : Y - p.off, Xo = p.merge ? X : X - p.off;
    TO* __restrict__ o = out + (((int64_t)tb * Ho + Yo) * Wo + Xo) * p.Cpad;
    if (p.merge) {
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        Elem<TO>::st(o + c, before[c]);
        Elem<TO>::st(o + 9 + c, warped[c]);
        Elem<TO>::st(o + 18 + c, hi[c]);
      }
      for (int c = 27; c < p.Cpad; ++c) Elem<TO>::st(o + c, 0.f);
    } else {
#pragma unroll
      for (int c = 0; c < 9; ++c) Elem<TO>::st(o + c, warped[c]);
      for (int c = 9; c < p.Cpad; ++c) Elem<TO>::st(o + c, 0.f);
    }
  }
}

// Backward of pack_d_fwd: the gradient of the discriminator's input w.r.t. the HR frames (tf.gradients through lib/Teco.py:224-272).
// Per HR pixel and triplet the forward kernel read: its own pixel of the three frames (merge mode), the centre frame once more inside
// the crop, and a bilinear footprint of the previous / next frame at the flow-displaced position -- the last one is a SCATTER here.
// Round 5: one wave = 64 consecutive HR columns x 4 rows, lane = column.  The element-per-thread form it replaces issued 36 fp32
// atomics per pixel (123 MB written for a 26 MB input, 133 us per step); now
//   * the footprint's right-hand contributions are handed to the lane holding the next column and its lower ones are kept for the next
//     row whenever the tap ADDRESSES line up (exact for any flow: a mismatch flushes them as they are) -- 3 instead of 12 atomics per
//     pixel and warped frame where the flow is smooth (the merged scatter of warp_s2d_bwd_kernel, warp.hip);
//   * the centre frame receives no scatter (the warped frames are the outer two): its two contributions are one plain read-modify-write.
template <typename TG>
__global__ __launch_bounds__(256) void pack_d_bwd_kernel(PackP p, float* __restrict__ d_frames) {
  constexpr int R = 4;
  const int H = 4 * p.h, W = 4 * p.w;
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const int xchunks = (W + 63) >> 6, bands = (H + R - 1) / R;
  const int64_t nwork = (int64_t)p.nt * p.B * bands * xchunks;
  const int64_t fsz = (int64_t)p.B * H * W * 3, flsz = (int64_t)p.B * p.h * p.w * 2;
  const int Ho = p.merge ? H : H - 2 * p.off, Wo = p.merge ? W : W - 2 * p.off;
  const TG* __restrict__ d_out = static_cast<const TG*>(p.out);
  const int wbase = p.merge ? 9 : 0;
  for (int64_t wi = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); wi < nwork; wi += (int64_t)gridDim.x * wpb) {
    const int xc = (int)(wi % xchunks), band = (int)((wi / xchunks) % bands);
    const int b = (int)((wi / ((int64_t)xchunks * bands)) % p.B), k = (int)(wi / ((int64_t)xchunks * bands * p.B));
    const int X = xc * 64 + lane;
    const bool xok = X < W;
    const int tb = k * p.B + b;
    const float* __restrict__ fl[2] = {p.flow_pre + p.idx_pre[k] * flsz, p.flow_nxt + p.idx_nxt[k] * flsz};
    float* __restrict__ dfr[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) dfr[t] = d_frames + (int64_t)(3 * k + t) * fsz + (int64_t)b * H * W * 3;
    // lower-row contributions of the previous row, per warped frame: left / right column of the footprint at offset paddr
    float pc[2][3], pd[2][3];
    int paddr[2] = {-1, -1};
    // horizontal hand-over + emission of one footprint row: `a` lands on offset addr, `bb` on addr + 3
    auto emit_row = [&](float* __restrict__ df, int addr, float (&a)[3], float (&bb)[3]) {
      const int right = __shfl_down(addr, 1, 64), left = __shfl_up(addr, 1, 64);
      const bool absorbed = addr >= 0 && right == addr + 3;          // the next lane's left column is my right column
      const bool takes = addr >= 0 && left >= 0 && left + 3 == addr;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float from_left = __shfl_up(bb[c], 1, 64);
        if (takes) a[c] += from_left;
      }
      if (addr >= 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (a[c] != 0.f) unsafeAtomicAdd(df + addr + c, a[c]);
          if (!absorbed && bb[c] != 0.f) unsafeAtomicAdd(df + addr + 3 + c, bb[c]);
        }
      }
    };
    for (int r = 0; r < R; ++r) {
      const int Y = band * R + r;
      if (Y >= H) break;                                               // wave-uniform
      const bool inside = xok && Y >= p.off && Y < H - p.off && X >= p.off && X < W - p.off;
      const bool live = xok && (p.merge || inside);
      const int Yo = p.merge ? Y : Y - p.off, Xo = p.merge ? X : X - p.off;
      const TG* __restrict__ g = d_out + (live ? (((int64_t)tb * Ho + Yo) * Wo + Xo) * p.Cpad : 0);
      const int own = (Y * W + X) * 3;
      // the pixel's own position: all three frames in merge mode, the centre frame once more inside the crop
      if (live) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float v1 = p.merge ? Elem<TG>::ld(g + c * 3 + 1) : 0.f;
          if (inside) v1 += Elem<TG>::ld(g + wbase + c * 3 + 1);
          dfr[1][own + c] += v1;                                       // no other thread of this launch touches the centre frame here
          if (p.merge) {
            unsafeAtomicAdd(dfr[0] + own + c, Elem<TG>::ld(g + c * 3 + 0));
            unsafeAtomicAdd(dfr[2] + own + c, Elem<TG>::ld(g + c * 3 + 2));
          }
        }
      }
      // upscale_four(4 flow) at (Y, X): the arithmetic of up4_flow (pack_d_fwd_kernel), corner loads per lane
      const int i = Y >> 2, j = min(X, W - 1) >> 2, i1 = min(i + 1, p.h - 1), j1 = min(j + 1, p.w - 1);
      const float wy = 0.25f * (Y & 3), wx = 0.25f * (X & 3);
      const float w0 = (1.f - wy) * (1.f - wx), w1 = (1.f - wy) * wx, w2 = wy * (1.f - wx), w3 = wy * wx;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const int t = 2 * w;
        const float2 c0 = *reinterpret_cast<const float2*>(fl[w] + ((int64_t)(b * p.h + i) * p.w + j) * 2);
        const float2 c1 = *reinterpret_cast<const float2*>(fl[w] + ((int64_t)(b * p.h + i) * p.w + j1) * 2);
        const float2 c2 = *reinterpret_cast<const float2*>(fl[w] + ((int64_t)(b * p.h + i1) * p.w + j) * 2);
        const float2 c3 = *reinterpret_cast<const float2*>(fl[w] + ((int64_t)(b * p.h + i1) * p.w + j1) * 2);
        const float fy = (c0.x * 4.f) * w0 + (c1.x * 4.f) * w1 + (c2.x * 4.f) * w2 + (c3.x * 4.f) * w3;
        const float fx = (c0.y * 4.f) * w0 + (c1.y * 4.f) * w1 + (c2.y * 4.f) * w2 + (c3.y * 4.f) * w3;
        const Tap2 tp = tap_of((float)Y - fy, (float)X - fx, H, W);
        const int addr = inside ? (tp.fy * W + tp.fx) * 3 : -1;
        float a[3], bb[3], cc[3], dd[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float gv = inside ? Elem<TG>::ld(g + wbase + c * 3 + t) : 0.f;
          a[c] = gv * (1.f - tp.ay) * (1.f - tp.ax);
          bb[c] = gv * (1.f - tp.ay) * tp.ax;
          cc[c] = gv * tp.ay * (1.f - tp.ax);
          dd[c] = gv * tp.ay * tp.ax;
        }
        // the previous row's lower contributions: they belong to this row's upper footprint row when the addresses agree
        if (paddr[w] >= 0) {
          if (paddr[w] == addr) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              a[c] += pc[w][c];
              bb[c] += pd[w][c];
            }
          } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              if (pc[w][c] != 0.f) unsafeAtomicAdd(dfr[t] + paddr[w] + c, pc[w][c]);
              if (pd[w][c] != 0.f) unsafeAtomicAdd(dfr[t] + paddr[w] + 3 + c, pd[w][c]);
            }
          }
        }
        emit_row(dfr[t], addr, a, bb);
        paddr[w] = addr >= 0 ? addr + W * 3 : -1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          pc[w][c] = cc[c];
          pd[w][c] = dd[c];
        }
      }
    }
    // the band's last lower rows
#pragma unroll
    for (int w = 0; w < 2; ++w) emit_row(dfr[2 * w], paddr[w], pc[w], pd[w]);
  }
}

static int fill_pack(PackP& p, const float* frames, const float* lr, const float* flow_pre, const float* flow_nxt,
                     const int* idx_pre, const int* idx_nxt, void* out, int B, int h, int w, int nt, int off, int merge,
                     int Cpad) {
  if (!(frames && flow_pre && flow_nxt && idx_pre && idx_nxt && out)) return 1;
  if (!(B > 0 && h > 0 && w > 0 && nt > 0 && nt <= 16 && off >= 0 && 2 * off < 4 * h && 2 * off < 4 * w)) return 1;
  if (merge ? !(lr && Cpad >= 27 && Cpad <= 32) : !(Cpad >= 9 && Cpad <= 32)) return 1;
  p.frames = frames; p.lr = lr; p.flow_pre = flow_pre; p.flow_nxt = flow_nxt; p.out = out;
  p.B = B; p.h = h; p.w = w; p.nt = nt; p.off = off; p.merge = merge; p.Cpad = Cpad;
  for (int k = 0; k < nt; ++k) {
    p.idx_pre[k] = idx_pre[k];
    p.idx_nxt[k] = idx_nxt[k];
  }
  return 0;
}

extern "C" int tg_pack_d_input_forward(const float* frames, const float* lr, const float* flow_pre,
                                       const float* flow_nxt, const int* idx_pre, const int* idx_nxt, void* out,
                                       int out_dtype, int B, int h, int w, int nt, int off, int merge, int Cpad,
                                       void* stream) {
  PackP p;
  TG_CHECK_ARG(fill_pack(p, frames, lr, flow_pre, flow_nxt, idx_pre, idx_nxt, out, B, h, w, nt, off, merge, Cpad) == 0,
               "bad argument");
  dim3 grid(grid_1d((int64_t)nt * B * h * w * 16, 256));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (out_dtype == TG_F32) hipLaunchKernelGGL((pack_d_fwd_kernel<float>), grid, dim3(256), 0, st, p);
  else if (out_dtype == TG_BF16) hipLaunchKernelGGL((pack_d_fwd_kernel<u16>), grid, dim3(256), 0, st, p);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

extern "C" int tg_pack_d_input_backward(const void* d_out, int dtype, const float* frames, const float* flow_pre,
                                        const float* flow_nxt, const int* idx_pre, const int* idx_nxt, float* d_frames,
                                        int B, int h, int w, int nt, int off, int merge, int Cpad, void* stream) {
  PackP p;
  TG_CHECK_ARG(d_frames != nullptr, "null d_frames");
  TG_CHECK_ARG(fill_pack(p, frames, frames, flow_pre, flow_nxt, idx_pre, idx_nxt, const_cast<void*>(d_out), B, h, w, nt,
                         off, merge, Cpad) == 0, "bad argument");
  // one wave per 4 rows x 64 columns of an image (pack_d_bwd_kernel)
  dim3 grid(grid_1d((int64_t)nt * B * ((4 * h + 3) / 4) * ((4 * w + 63) / 64), 4));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == TG_F32) hipLaunchKernelGGL((pack_d_bwd_kernel<float>), TG_DET_GRID(grid), TG_DET_WAVE(256), 0, st, p, d_frames);
  else if (dtype == TG_BF16) hipLaunchKernelGGL((pack_d_bwd_kernel<u16>), TG_DET_GRID(grid), TG_DET_WAVE(256), 0, st, p, d_frames);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}
