// Backward-warp kernels (gfx950, HBM/latency bound, wave64 shuffles).
//
//  tg_warp_s2d_*  fuses the recurrent input builder of the FRVSR/TecoGAN step
//     flow_hr = upscale_four(4*flow_lr)                 reference lib/ops.py:126-163, lib/Teco.py:113, main.py:213
//     warped  = dense_image_warp(pre_hr, flow_hr)       lib/Teco.py:140, main.py:215   [TF1] SURVEY A.5
//     x       = deprocess(warped)                       lib/Teco.py:143
//     s2d     = space_to_depth(x, 4)                    lib/Teco.py:145-148, main.py:201   (bit-exact shuffle)
//     gen_in  = concat(LR frame, s2d)                   lib/Teco.py:150, main.py:202
//  in one pass: per frame it reads the previous HR frame + the LR flow and writes the 51(+pad)-channel
//  generator input; flow_hr and the warped HR frame are never materialised.
//  Thread map: 16 consecutive lanes own one LR pixel (lane&15 = dy*4+dx), so the 48 s2d channels of
//  a pixel are written as 16 consecutive 12-byte pieces (coalesced) and the 4 LR flow corners are
//  loaded once by lanes 0..3 of the group and broadcast with __shfl.
//  tg_warp_* is the plain dense_image_warp used for the LR warp loss and the discriminator inputs.
#include "common.h"
#include <stdlib.h>

struct BilinearTap {
  int fy, fx;        // clamped floor
  float ay, ax;      // clamped alpha
  bool gy, gx;       // gradient passes to the query (0 < alpha_raw <= 1)   [TF1]
};

__device__ __forceinline__ BilinearTap make_tap(float qy, float qx, int H, int W) {
  BilinearTap t;
  const float fy = fminf(fmaxf(floorf(qy), 0.f), (float)(H - 2));
  const float fx = fminf(fmaxf(floorf(qx), 0.f), (float)(W - 2));
  const float ry = qy - fy, rx = qx - fx;
  t.fy = (int)fy;
  t.fx = (int)fx;
  t.ay = fminf(fmaxf(ry, 0.f), 1.f);
  t.ax = fminf(fmaxf(rx, 0.f), 1.f);
  t.gy = ry > 0.f && ry <= 1.f;
  t.gx = rx > 0.f && rx <= 1.f;
  return t;
}

__device__ __forceinline__ int mirror(int i, int n) { return i < n ? i : 2 * n - 1 - i; }  // SYMMETRIC pad

// upscale_four(4*flow_lr) at HR position (4i+dy, 4j+dx); corners broadcast within the 16-lane group.
__device__ __forceinline__ float2 flow_hr_at(const float* __restrict__ flow_lr, int b, int i, int j, int sub,
                                             int h, int w, int hf, int wf, float* wts /*[4] out*/) {
  const int dy = sub >> 2, dx = sub & 3;
  const int i1 = min(i + 1, h - 1), j1 = min(j + 1, w - 1);
  float2 mine = make_float2(0.f, 0.f);
  if (sub < 4) {
    const int ci = mirror((sub & 2) ? i1 : i, hf), cj = mirror((sub & 1) ? j1 : j, wf);
    mine = *reinterpret_cast<const float2*>(flow_lr + ((int64_t)(b * hf + ci) * wf + cj) * 2);
  }
  float2 c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c[k].x = __shfl(mine.x, k, 16);
    c[k].y = __shfl(mine.y, k, 16);
  }
  const float wy = 0.25f * dy, wx = 0.25f * dx;
  wts[0] = (1.f - wy) * (1.f - wx);
  wts[1] = (1.f - wy) * wx;
  wts[2] = wy * (1.f - wx);
  wts[3] = wy * wx;
  float2 f;
  // gen_flow = upscale_four(gen_flow_lr * 4.0): scale first, then blend (lib/Teco.py:113)
  f.x = (c[0].x * 4.f) * wts[0] + (c[1].x * 4.f) * wts[1] + (c[2].x * 4.f) * wts[2] + (c[3].x * 4.f) * wts[3];
  f.y = (c[0].y * 4.f) * wts[0] + (c[1].y * 4.f) * wts[1] + (c[2].y * 4.f) * wts[2] + (c[3].y * 4.f) * wts[3];
  return f;
}

template <typename TOut>
__global__ __launch_bounds__(256) void warp_s2d_fwd_scalar_kernel(const float* __restrict__ pre,
                                                           const float* __restrict__ flow_lr,
                                                           const float* __restrict__ lr, TOut* __restrict__ out,
                                                           int B, int h, int w, int hf, int wf, int Cpad, float scale,
                                                           float shift, float* __restrict__ warped) {
  const int64_t npix = (int64_t)B * h * w;
  const int H = 4 * h, W = 4 * w;
  for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; (gid >> 4) < npix;
       gid += (int64_t)gridDim.x * blockDim.x) {
    const int64_t lp = gid >> 4;
    const int sub = (int)(gid & 15);
    const int j = (int)(lp % w);
    const int i = (int)((lp / w) % h);
    const int b = (int)(lp / ((int64_t)w * h));
    TOut* __restrict__ o = out + lp * Cpad;
    if (sub < 3) Elem<TOut>::st(o + sub, lr[lp * 3 + sub]);
    if (51 + sub < Cpad) Elem<TOut>::st(o + 51 + sub, 0.f);
    float v[3] = {0.f, 0.f, 0.f};
    if (pre) {
      float wts[4];
      const float2 f = flow_hr_at(flow_lr, b, i, j, sub, h, w, hf, wf, wts);
      const int Y = 4 * i + (sub >> 2), X = 4 * j + (sub & 3);
      const BilinearTap t = make_tap((float)Y - f.x, (float)X - f.y, H, W);
      const float* __restrict__ p0 = pre + ((int64_t)(b * H + t.fy) * W + t.fx) * 3;
      const float* __restrict__ p1 = p0 + (int64_t)W * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float tl = p0[c], tr = p0[3 + c], bl = p1[c], br = p1[3 + c];
        const float top = t.ax * (tr - tl) + tl;
        const float bot = t.ax * (br - bl) + bl;
        const float wv = t.ay * (bot - top) + top;
        if (warped) warped[((int64_t)(b * H + Y) * W + X) * 3 + c] = wv;
        v[c] = wv * scale + shift;
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) Elem<TOut>::st(o + 3 + sub * 3 + c, v[c]);
  }
}


// Row-band form (the scalar kernel above is the fallback for odd channel paddings).  One wave owns 16 consecutive LR pixels of
// one LR row = a 4 x 64 band of HR pixels; a workgroup = 4 vertically adjacent bands (17 HR rows of the previous frame for 16
// produced).  Lane l is HR column 64*jt + l, so (a) the 2x2 bilinear footprints of a wave -- two 24-byte row segments per
// lane (tl|tr, bl|br: 6 consecutive floats, dwordx4 + dwordx2) -- cover one ~780-byte stretch per HR row instead of four
// 60-byte ones, (b) the four HR rows of the band are independent: their flows come from ONE set of four LR corner loads per
// lane and all eight footprint loads are in flight together (the 16-lanes-per-LR-pixel form this replaces made four dependent
// round trips per wave at 1080p: 15.2 us = 2.76 TB/s, latency-bound), and (c) the generator-input rows of the 16 LR pixels
// (Cpad channels each: LR frame | 48 space-to-depth channels | zero padding) are assembled in LDS and leave as ONE contiguous
// run of 16-byte vectors (1792 bytes per wave at bf16, Cpad 56).  The LDS executes one wave's operations in order, so the
// hand-off needs no workgroup barrier.  Workgroups are renumbered so that each XCD (workgroup id mod 8) owns a contiguous
// range of bands and shared footprint rows meet in one L2.
struct __attribute__((packed, aligned(4))) F6 { float v[6]; };

template <typename TOut>
__global__ __launch_bounds__(256) void warp_s2d_fwd_kernel(const float* __restrict__ pre,
                                                           const float* __restrict__ flow_lr,
                                                           const float* __restrict__ lr, TOut* __restrict__ out,
                                                           int B, int h, int w, int hf, int wf, int Cpad, float scale,
                                                           float shift, float* __restrict__ warped) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 4 waves x 16 rows of Cpad elements
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jl = lane >> 2, xs = lane & 3;
  const int tw = (w + 15) >> 4, th = (h + 3) >> 2;
  // XCD-contiguous renumbering (a bijection for any grid size)
  const int nb = (int)gridDim.x, xcd = (int)blockIdx.x & 7;
  int bid = xcd * (nb >> 3) + min(xcd, nb & 7) + ((int)blockIdx.x >> 3);
  const int jt = bid % tw;
  bid /= tw;
  const int i = (bid % th) * 4 + wave, b = bid / th;
  if (i >= h) return;                                        // wave-uniform; the kernel has no workgroup barrier
  const int j0 = jt * 16;
  const int nv = min(16, w - j0);                            // LR pixels of this band that exist
  const int j = min(j0 + jl, w - 1);                         // lanes past the row end redo its last pixel (never stored)
  const bool live = j0 + jl < w;
  const int64_t lp0 = ((int64_t)b * h + i) * w + j0;
  TOut* __restrict__ rows = reinterpret_cast<TOut*>(smem) + wave * 16 * Cpad;
  const float lrv = lr[lp0 * 3 + min(lane, nv * 3 - 1)];     // unconditional: issued here, first used after the footprint loads
  float v[4][3];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r][0] = v[r][1] = v[r][2] = 0.f;
  if (pre) {
    const int H = 4 * h, W = 4 * w;
    const int i1 = min(i + 1, h - 1), j1 = min(j + 1, w - 1);
    float2 c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ci = mirror((k & 2) ? i1 : i, hf), cj = mirror((k & 1) ? j1 : j, wf);
      c[k] = *reinterpret_cast<const float2*>(flow_lr + ((int64_t)(b * hf + ci) * wf + cj) * 2);
    }
    const int X = 4 * j + xs;
    BilinearTap t[4];
    F6 top6[4], bot6[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // gen_flow = upscale_four(gen_flow_lr * 4.0): scale first, then blend (lib/Teco.py:113) -- the arithmetic of flow_hr_at
      const float wy = 0.25f * r, wx = 0.25f * xs;
      const float w0 = (1.f - wy) * (1.f - wx), w1 = (1.f - wy) * wx, w2 = wy * (1.f - wx), w3 = wy * wx;
      const float fy = (c[0].x * 4.f) * w0 + (c[1].x * 4.f) * w1 + (c[2].x * 4.f) * w2 + (c[3].x * 4.f) * w3;
      const float fx = (c[0].y * 4.f) * w0 + (c[1].y * 4.f) * w1 + (c[2].y * 4.f) * w2 + (c[3].y * 4.f) * w3;
      t[r] = make_tap((float)(4 * i + r) - fy, (float)X - fx, H, W);
      const float* __restrict__ p0 = pre + ((int64_t)(b * H + t[r].fy) * W + t[r].fx) * 3;
      top6[r] = *reinterpret_cast<const F6*>(p0);
      bot6[r] = *reinterpret_cast<const F6*>(p0 + (int64_t)W * 3);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float tl = top6[r].v[ch], tr = top6[r].v[3 + ch], bl = bot6[r].v[ch], br = bot6[r].v[3 + ch];
        const float top = t[r].ax * (tr - tl) + tl;
        const float bot = t[r].ax * (br - bl) + bl;
        const float wv = t[r].ay * (bot - top) + top;
        if (warped && live) warped[((int64_t)(b * H + 4 * i + r) * W + X) * 3 + ch] = wv;
        v[r][ch] = wv * scale + shift;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) Elem<TOut>::st(rows + jl * Cpad + 3 + (r * 4 + xs) * 3 + ch, v[r][ch]);
  if (lane < nv * 3) Elem<TOut>::st(rows + (lane / 3) * Cpad + lane % 3, lrv);
  for (int k = 51 + xs; k < Cpad; k += 4) Elem<TOut>::st(rows + jl * Cpad + k, 0.f);          // zero padding of row jl
  __builtin_amdgcn_wave_barrier();                           // same-wave LDS ops are ordered; keep the compiler from mixing them
  const int nvec = nv * Cpad * (int)sizeof(TOut) / 16;
  const unsigned char* __restrict__ src = smem + (size_t)wave * 16 * Cpad * sizeof(TOut);
  unsigned char* __restrict__ dst = reinterpret_cast<unsigned char*>(out) + lp0 * Cpad * (int64_t)sizeof(TOut);
  for (int e = lane; e < nvec; e += 64)
    *reinterpret_cast<uint4*>(dst + e * 16) = *reinterpret_cast<const uint4*>(src + e * 16);
}

// Backward of the fused warp: gather of d_out, scatter into d_pre (the previous HR frame's gradient) and into the LR flow.
// The scatter is the cost (12 fp32 atomics per HR pixel: 16.5 us per [4,128,128] frame, one launch per BPTT frame on the
// critical path).  MERGE: where the flow is smooth the footprints of neighbouring HR pixels tile the frame -- the right tap of
// pixel X IS the left tap of pixel X+1, the lower taps of row Y the upper taps of row Y+1.  Lanes therefore hand their
// tr / br contributions to the lane on their right and their (merged) bl / br to the lane below whenever the ADDRESSES say
// so (o00_left + 3 == o00_right, o00_up + 3W == o00_down: the test is on addresses only, so it is exact for any flow), and a
// contribution that has been handed on (or is zero) issues no atomic: 3 instead of 12 atomics per pixel in smooth regions,
// the old count where neighbouring footprints do not line up.  Lane pairs: left/right along the HR row incl. the step into the
// next LR pixel of the wave (lane +-1 / +-13), up/down inside the 4x4 block of an LR pixel (lane +-4).
template <typename TG>
__global__ __launch_bounds__(256) void warp_s2d_bwd_kernel(const TG* __restrict__ d_out, const float* __restrict__ pre,
                                                           const float* __restrict__ flow_lr,
                                                           float* __restrict__ d_pre, float* __restrict__ d_flow_lr,
                                                           int B, int h, int w, int Cpad, float scale, int merge) {
  const int64_t npix = (int64_t)B * h * w, nlane = npix * 16;
  const int H = 4 * h, W = 4 * w, W3 = W * 3;
  const int lane = threadIdx.x & 63, sub = lane & 15, dy = sub >> 2, dx = sub & 3;
  const int lsrc = dx > 0 ? lane - 1 : lane - 13, rdst = dx < 3 ? lane + 1 : lane + 13;
  const bool has_l = lsrc >= 0, has_r = rdst < 64, has_u = dy > 0, has_d = dy < 3;
  const int lfrom = has_l ? lsrc : lane, rfrom = has_r ? rdst : lane, ufrom = has_u ? lane - 4 : lane, dfrom = has_d ? lane + 4 : lane;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < nlane;
       base += (int64_t)gridDim.x * blockDim.x) {                                                   // wave-uniform trip count
    const int64_t gid = base + lane;
    const bool active = gid < nlane;                         // whole 16-lane groups past the end idle (zero contributions)
    const int64_t lp = active ? (gid >> 4) : npix - 1;
    const int j = (int)(lp % w);
    const int i = (int)((lp / w) % h);
    const int b = (int)(lp / ((int64_t)w * h));
    float wts[4];
    const float2 f = flow_hr_at(flow_lr, b, i, j, sub, h, w, h, w, wts);
    const int Y = 4 * i + dy, X = 4 * j + dx;
    const BilinearTap t = make_tap((float)Y - f.x, (float)X - f.y, H, W);
    const int64_t o00 = ((int64_t)(b * H + t.fy) * W + t.fx) * 3;
    const int64_t o10 = o00 + W3;
    float d_ax = 0.f, d_ay = 0.f;
    float ctl[3], ctr[3], cbl[3], cbr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float g = active ? Elem<TG>::ld(d_out + lp * Cpad + 3 + sub * 3 + c) * scale : 0.f;
      const float tl = pre[o00 + c], tr = pre[o00 + 3 + c], bl = pre[o10 + c], br = pre[o10 + 3 + c];
      const float top = t.ax * (tr - tl) + tl;
      const float bot = t.ax * (br - bl) + bl;
      d_ay += g * (bot - top);
      d_ax += g * ((1.f - t.ay) * (tr - tl) + t.ay * (br - bl));
      ctl[c] = g * (1.f - t.ay) * (1.f - t.ax);
      ctr[c] = g * (1.f - t.ay) * t.ax;
      cbl[c] = g * t.ay * (1.f - t.ax);
      cbr[c] = g * t.ay * t.ax;
    }
    if (d_pre) {
      if (merge) {                                           // kernel argument: uniform
        const int o = active ? (int)o00 : -8;                // (the launcher checks that the frame has < 2^31 elements)
        const int oL = __shfl(o, lfrom), oR = __shfl(o, rfrom);
        const bool recv_l = has_l && oL + 3 == o, send_r = has_r && o + 3 == oR;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float a = __shfl(ctr[c], lfrom), q = __shfl(cbr[c], lfrom);
          if (recv_l) { ctl[c] += a; cbl[c] += q; }
        }
        if (send_r) {
#pragma unroll
          for (int c = 0; c < 3; ++c) ctr[c] = cbr[c] = 0.f;
        }
        const int oU = __shfl(o, ufrom), oD = __shfl(o, dfrom);
        const bool recv_u = has_u && oU + W3 == o, send_d = has_d && o + W3 == oD;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float a = __shfl(cbl[c], ufrom), q = __shfl(cbr[c], ufrom);
          if (recv_u) { ctl[c] += a; ctr[c] += q; }
        }
        if (send_d) {
#pragma unroll
          for (int c = 0; c < 3; ++c) cbl[c] = cbr[c] = 0.f;
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (ctl[c] != 0.f) unsafeAtomicAdd(d_pre + o00 + c, ctl[c]);
        if (ctr[c] != 0.f) unsafeAtomicAdd(d_pre + o00 + 3 + c, ctr[c]);
        if (cbl[c] != 0.f) unsafeAtomicAdd(d_pre + o10 + c, cbl[c]);
        if (cbr[c] != 0.f) unsafeAtomicAdd(d_pre + o10 + 3 + c, cbr[c]);
      }
    }
    if (d_flow_lr) {
      // q = pos - flow  ->  dflow = -dq ; flow_hr = sum_k wts[k] * 4 * corner_k
      const float dfy = t.gy ? -d_ay * 4.f : 0.f, dfx = t.gx ? -d_ax * 4.f : 0.f;
      float cy[4], cx[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        cy[k] = dfy * wts[k];
        cx[k] = dfx * wts[k];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          cy[k] += __shfl_xor(cy[k], o, 16);
          cx[k] += __shfl_xor(cx[k], o, 16);
        }
      }
      if (sub == 0 && active) {
        const int i1 = min(i + 1, h - 1), j1 = min(j + 1, w - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int ci = (k & 2) ? i1 : i, cj = (k & 1) ? j1 : j;
          float* dst = d_flow_lr + ((int64_t)(b * h + ci) * w + cj) * 2;
          unsafeAtomicAdd(dst, cy[k]);
          unsafeAtomicAdd(dst + 1, cx[k]);
        }
      }
    }
  }
}

extern "C" int tg_warp_s2d_forward(const float* pre, const float* flow_lr, const float* lr, void* out, int out_dtype,
                                   int B, int h, int w, int hf, int wf, int Cpad, float scale, float shift,
                                   float* warped, void* stream) {
  TG_CHECK_ARG(lr && out, "null pointer");
  TG_CHECK_ARG(pre == nullptr || flow_lr != nullptr, "flow_lr required with pre");
  TG_CHECK_ARG(B > 0 && h > 0 && w > 0 && Cpad >= 51 && Cpad <= 67, "bad shape (51 <= Cpad <= 67)");
  TG_CHECK_ARG(hf > 0 && wf > 0 && hf <= h && wf <= w && 2 * hf >= h && 2 * wf >= w, "bad flow extent");
  const int64_t work = (int64_t)B * h * w * 16;
  int grid = grid_1d(work, 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // algorithmic bytes: previous HR frame + LR flow + LR frame read once, generator input written once
  const double px = (double)B * h * w;
  const double by = px * ((pre ? 16.0 * 12.0 : 0.0) + (pre ? (double)hf * wf / ((double)h * w) * 8.0 : 0.0) + 12.0 +
                          Cpad * (out_dtype == TG_F32 ? 4.0 : 2.0));
  const int esz = out_dtype == TG_F32 ? 4 : 2;
  const int64_t bands = (int64_t)B * ((h + 3) / 4) * ((w + 15) / 16);       // one workgroup per 4 LR rows x 16 LR pixels
  const bool vec = (Cpad * esz) % 16 == 0 && Cpad * esz <= 256 && ((uintptr_t)out & 15) == 0 && bands < ((int64_t)1 << 31);
  const unsigned lds = 64u * Cpad * esz;
  if (vec) grid = (int)bands;
  if (out_dtype == TG_F32 && vec)
    TG_LAUNCH("warp_s2d_fwd<f32>", 0, by, (warp_s2d_fwd_kernel<float>), dim3(grid), dim3(256), lds, st, pre, flow_lr, lr,
              (float*)out, B, h, w, hf, wf, Cpad, scale, shift, warped);
  else if (out_dtype == TG_BF16 && vec)
    TG_LAUNCH("warp_s2d_fwd<bf16>", 0, by, (warp_s2d_fwd_kernel<u16>), dim3(grid), dim3(256), lds, st, pre, flow_lr, lr,
              (u16*)out, B, h, w, hf, wf, Cpad, scale, shift, warped);
  else if (out_dtype == TG_F32)
    TG_LAUNCH("warp_s2d_fwd_scalar<f32>", 0, by, (warp_s2d_fwd_scalar_kernel<float>), dim3(grid), dim3(256), 0, st, pre,
              flow_lr, lr, (float*)out, B, h, w, hf, wf, Cpad, scale, shift, warped);
  else if (out_dtype == TG_BF16)
    TG_LAUNCH("warp_s2d_fwd_scalar<bf16>", 0, by, (warp_s2d_fwd_scalar_kernel<u16>), dim3(grid), dim3(256), 0, st, pre,
              flow_lr, lr, (u16*)out, B, h, w, hf, wf, Cpad, scale, shift, warped);
  else
    TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

extern "C" int tg_warp_s2d_backward(const void* d_out, int dtype, const float* pre, const float* flow_lr, float* d_pre,
                                    float* d_flow_lr, int B, int h, int w, int Cpad, float scale, void* stream) {
  TG_CHECK_ARG(d_out && pre && flow_lr, "null pointer");
  TG_CHECK_ARG(B > 0 && h > 0 && w > 0 && Cpad >= 51, "bad shape");
  const int64_t work = (int64_t)B * h * w * 16;
  int grid = grid_1d(work, 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double by = (double)B * h * w * (Cpad * (dtype == TG_F32 ? 4.0 : 2.0) + 16.0 * 12.0 * 3.0 + 16.0);
  const int merge = (int64_t)B * h * w * 48 + (int64_t)w * 12 + 16 < ((int64_t)1 << 31);   // 32-bit tap offsets
  if (dtype == TG_F32)
    TG_LAUNCH("warp_s2d_bwd<f32>", 0, by, (warp_s2d_bwd_kernel<float>), TG_DET_GRID(grid), TG_DET_WAVE(256), 0, st, (const float*)d_out, pre,
              flow_lr, d_pre, d_flow_lr, B, h, w, Cpad, scale, merge);
  else if (dtype == TG_BF16)
    TG_LAUNCH("warp_s2d_bwd<bf16>", 0, by, (warp_s2d_bwd_kernel<u16>), TG_DET_GRID(grid), TG_DET_WAVE(256), 0, st, (const u16*)d_out, pre,
              flow_lr, d_pre, d_flow_lr, B, h, w, Cpad, scale, merge);
  else
    TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// plain dense_image_warp
__global__ __launch_bounds__(256) void warp_fwd_kernel(const float* __restrict__ img, const float* __restrict__ flow,
                                                       float* __restrict__ out, int B, int H, int W, int C) {
  const int64_t n = (int64_t)B * H * W;
  for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < n; pix += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    const float2 f = *reinterpret_cast<const float2*>(flow + pix * 2);
    const BilinearTap t = make_tap((float)y - f.x, (float)x - f.y, H, W);
    const float* __restrict__ p0 = img + ((int64_t)(b * H + t.fy) * W + t.fx) * C;
    const float* __restrict__ p1 = p0 + (int64_t)W * C;
    for (int c = 0; c < C; ++c) {
      const float tl = p0[c], tr = p0[C + c], bl = p1[c], br = p1[C + c];
      const float top = t.ax * (tr - tl) + tl;
      const float bot = t.ax * (br - bl) + bl;
      out[pix * C + c] = t.ay * (bot - top) + top;
    }
  }
}

__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ img,
                                                       const float* __restrict__ flow, float* __restrict__ d_img,
                                                       float* __restrict__ d_flow, int B, int H, int W, int C) {
  const int64_t n = (int64_t)B * H * W;
  for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < n; pix += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    const float2 f = *reinterpret_cast<const float2*>(flow + pix * 2);
    const BilinearTap t = make_tap((float)y - f.x, (float)x - f.y, H, W);
    const int64_t o00 = ((int64_t)(b * H + t.fy) * W + t.fx) * C;
    const int64_t o10 = o00 + (int64_t)W * C;
    float d_ax = 0.f, d_ay = 0.f;
    for (int c = 0; c < C; ++c) {
      const float g = d_out[pix * C + c];
      const float tl = img[o00 + c], tr = img[o00 + C + c], bl = img[o10 + c], br = img[o10 + C + c];
      const float top = t.ax * (tr - tl) + tl;
      const float bot = t.ax * (br - bl) + bl;
      d_ay += g * (bot - top);
      d_ax += g * ((1.f - t.ay) * (tr - tl) + t.ay * (br - bl));
      if (d_img) {
        unsafeAtomicAdd(d_img + o00 + c, g * (1.f - t.ay) * (1.f - t.ax));
        unsafeAtomicAdd(d_img + o00 + C + c, g * (1.f - t.ay) * t.ax);
        unsafeAtomicAdd(d_img + o10 + c, g * t.ay * (1.f - t.ax));
        unsafeAtomicAdd(d_img + o10 + C + c, g * t.ay * t.ax);
      }
    }
    if (d_flow) {
      d_flow[pix * 2] = t.gy ? -d_ay : 0.f;
      d_flow[pix * 2 + 1] = t.gx ? -d_ax : 0.f;
    }
  }
}

extern "C" int tg_warp_forward(const float* img, const float* flow, float* out, int B, int H, int W, int C,
                               void* stream) {
  TG_CHECK_ARG(img && flow && out && B > 0 && H > 1 && W > 1 && C > 0, "bad argument");
  const int grid = grid_1d((int64_t)B * H * W, 256);
  hipLaunchKernelGGL(warp_fwd_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), img, flow, out, B, H,
                     W, C);
  TG_CHECK_LAUNCH();
}

extern "C" int tg_warp_backward(const float* d_out, const float* img, const float* flow, float* d_img, float* d_flow,
                                int B, int H, int W, int C, void* stream) {
  TG_CHECK_ARG(d_out && img && flow && B > 0 && H > 1 && W > 1 && C > 0, "bad argument");
  const int grid = grid_1d((int64_t)B * H * W, 256);
  hipLaunchKernelGGL(warp_bwd_kernel, TG_DET_GRID(grid), TG_DET_WAVE(256), 0, static_cast<hipStream_t>(stream), d_out, img, flow,
                     d_img, d_flow, B, H, W, C);
  TG_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// upscale_four / legacy bilinear x4: out = gain * up4(in)
__global__ __launch_bounds__(256) void upscale4_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int B, int h, int w, int C, float gain) {
  const int H = 4 * h, W = 4 * w;
  const int64_t n = (int64_t)B * H * W * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    int64_t t = e / C;
    const int X = (int)(t % W);
    t /= W;
    const int Y = (int)(t % H), b = (int)(t / H);
    const int i = Y >> 2, j = X >> 2, i1 = min(i + 1, h - 1), j1 = min(j + 1, w - 1);
    const float wy = 0.25f * (Y & 3), wx = 0.25f * (X & 3);
    const float* __restrict__ base = in + (int64_t)b * h * w * C + c;
    const float tl = base[((int64_t)i * w + j) * C] * gain, tr = base[((int64_t)i * w + j1) * C] * gain;
    const float bl = base[((int64_t)i1 * w + j) * C] * gain, br = base[((int64_t)i1 * w + j1) * C] * gain;
    out[e] = tl * (1.f - wy) * (1.f - wx) + tr * (1.f - wy) * wx + bl * wy * (1.f - wx) + br * wy * wx;
  }
}

__global__ __launch_bounds__(256) void upscale4_bwd_kernel(const float* __restrict__ d_out, float* __restrict__ d_in,
                                                           int B, int h, int w, int C, float gain) {
  // gather form: LR element (i,j) collects from the HR blocks for which it is a corner
  const int H = 4 * h, W = 4 * w;
  const int64_t n = (int64_t)B * h * w * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    int64_t t = e / C;
    const int j = (int)(t % w);
    t /= w;
    const int i = (int)(t % h), b = (int)(t / h);
    const float* __restrict__ g = d_out + (int64_t)b * H * W * C + c;
    float s = 0.f;
    for (int bi = i - 1; bi <= i; ++bi) {
      if (bi < 0) continue;
      for (int bj = j - 1; bj <= j; ++bj) {
        if (bj < 0) continue;
        const int bi1 = min(bi + 1, h - 1), bj1 = min(bj + 1, w - 1);
        for (int dy = 0; dy < 4; ++dy)
          for (int dx = 0; dx < 4; ++dx) {
            const float wy = 0.25f * dy, wx = 0.25f * dx;
            float wgt = 0.f;
            if (bi == i && bj == j) wgt += (1.f - wy) * (1.f - wx);
            if (bi == i && bj1 == j) wgt += (1.f - wy) * wx;
            if (bi1 == i && bj == j) wgt += wy * (1.f - wx);
            if (bi1 == i && bj1 == j) wgt += wy * wx;
            if (wgt != 0.f) s += wgt * g[((int64_t)(4 * bi + dy) * W + 4 * bj + dx) * C];
          }
      }
    }
    d_in[e] = s * gain;
  }
}

extern "C" int tg_upscale4_forward(const float* in, float* out, int B, int h, int w, int C, float gain, void* stream) {
  TG_CHECK_ARG(in && out && B > 0 && h > 0 && w > 0 && C > 0, "bad argument");
  const int grid = grid_1d((int64_t)B * h * w * 16 * C, 256);
  hipLaunchKernelGGL(upscale4_fwd_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, B, h, w,
                     C, gain);
  TG_CHECK_LAUNCH();
}

extern "C" int tg_upscale4_backward(const float* d_out, float* d_in, int B, int h, int w, int C, float gain,
                                    void* stream) {
  TG_CHECK_ARG(d_out && d_in && B > 0 && h > 0 && w > 0 && C > 0, "bad argument");
  const int grid = grid_1d((int64_t)B * h * w * C, 256);
  hipLaunchKernelGGL(upscale4_bwd_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), d_out, d_in, B, h,
                     w, C, gain);
  TG_CHECK_LAUNCH();
}
