// HBM-bound companions of the convolution engine (gfx950): pooling, legacy resizes, bicubic
// epilogue, BatchNorm+LeakyReLU, activation gradients, TF-flavoured Adam, weight re-layout and the
// loss reductions.  Each kernel cites the reference op it replaces.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// slim.max_pool2d([2,2]) stride 2 VALID -- reference lib/ops.py:92-93  [TF1] A.3
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H,
                                                           int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t n = (int64_t)N * Ho * Wo * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    int64_t t = e / C;
    const int x = (int)(t % Wo);
    t /= Wo;
    const int y = (int)(t % Ho), b = (int)(t / Ho);
    const T* __restrict__ p = in + ((int64_t)(b * H + 2 * y) * W + 2 * x) * C + c;
    const float v0 = Elem<T>::ld(p), v1 = Elem<T>::ld(p + C), v2 = Elem<T>::ld(p + (int64_t)W * C),
                v3 = Elem<T>::ld(p + (int64_t)W * C + C);
    Elem<T>::st(out + e, fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ in, const T* __restrict__ d_out,
                                                           T* __restrict__ d_in, int N, int H, int W, int C, int act,
                                                           float alpha, const T* __restrict__ add) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t n = (int64_t)N * H * W * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    int64_t t = e / C;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H), b = (int)(t / H);
    float g = 0.f;
    const int oy = y >> 1, ox = x >> 1;
    // `add`: a second gradient w.r.t. the same (post-activation) tensor, e.g. a VGG feature tap (lib/Teco.py:346-352)
    if (add) g = Elem<T>::ld(add + e) * act_grad_from_out(Elem<T>::ld(in + e), act, alpha);
    if (oy < Ho && ox < Wo) {
      const T* __restrict__ p = in + ((int64_t)(b * H + 2 * oy) * W + 2 * ox) * C + c;
      const float v[4] = {Elem<T>::ld(p), Elem<T>::ld(p + C), Elem<T>::ld(p + (int64_t)W * C),
                          Elem<T>::ld(p + (int64_t)W * C + C)};
      int am = 0;
      float m = v[0];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k] > m) {
          m = v[k];
          am = k;
        }
      if (am == ((y & 1) * 2 + (x & 1)))  // `in` is the activation OUTPUT: fuse its derivative here
        g += Elem<T>::ld(d_out + ((int64_t)(b * Ho + oy) * Wo + ox) * C + c) * act_grad_from_out(m, act, alpha);
    }
    Elem<T>::st(d_in + e, g);
  }
}

// bf16, C % 8 == 0: one thread per (output pixel, channel octet), 16-byte loads and stores, 32-bit index math.
__global__ __launch_bounds__(256) void maxpool2_fwd_x8_kernel(const u16* __restrict__ in, u16* __restrict__ out, int N,
                                                              int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, C8 = C >> 3;
  const int n = N * Ho * Wo * C8;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int c8 = e % C8;
    int t = e / C8;
    const int x = t % Wo;
    t /= Wo;
    const int y = t % Ho, b = t / Ho;
    const u16* __restrict__ p = in + ((int64_t)(b * H + 2 * y) * W + 2 * x) * C + c8 * 8;
    const uint4 q0 = *reinterpret_cast<const uint4*>(p), q1 = *reinterpret_cast<const uint4*>(p + C);
    const uint4 q2 = *reinterpret_cast<const uint4*>(p + (int64_t)W * C), q3 = *reinterpret_cast<const uint4*>(p + (int64_t)W * C + C);
    float v0[8], v1[8], v2[8], v3[8], m[8];
    bf8_unpack(q0, v0); bf8_unpack(q1, v1); bf8_unpack(q2, v2); bf8_unpack(q3, v3);
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = fmaxf(fmaxf(v0[k], v1[k]), fmaxf(v2[k], v3[k]));
    *reinterpret_cast<uint4*>(out + (int64_t)e * 8) = bf8_pack(m);
  }
}

// One thread per (2x2 window, channel octet): reads the window and the pooled gradient once, writes all four input
// gradients (first-max-wins tie rule of the scalar kernel; odd trailing rows/columns are zeroed by the last windows).
__global__ __launch_bounds__(256) void maxpool2_bwd_x8_kernel(const u16* __restrict__ in, const u16* __restrict__ d_out,
                                                              u16* __restrict__ d_in, int N, int H, int W, int C, int act,
                                                              float alpha, const u16* __restrict__ add) {
  const int Ho = H / 2, Wo = W / 2, C8 = C >> 3;
  const int Hc = (H + 1) / 2, Wc = (W + 1) / 2;                // windows incl. the ragged edge
  const int n = N * Hc * Wc * C8;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int c8 = e % C8;
    int t = e / C8;
    const int ox = t % Wc;
    t /= Wc;
    const int oy = t % Hc, b = t / Hc;
    const int64_t base = ((int64_t)(b * H + 2 * oy) * W + 2 * ox) * C + c8 * 8;
    // gradient arriving directly at this pixel (`add`, nullable) times the activation derivative
    auto direct = [&](int64_t off) -> uint4 {
      if (!add) return make_uint4(0, 0, 0, 0);
      float a[8], xv[8];
      bf8_unpack(*reinterpret_cast<const uint4*>(add + off), a);
      bf8_unpack(*reinterpret_cast<const uint4*>(in + off), xv);
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] *= act_grad_from_out(xv[k], act, alpha);
      return bf8_pack(a);
    };
    if (oy >= Ho || ox >= Wo) {                                 // ragged edge: no pooling window covers these pixels
      *reinterpret_cast<uint4*>(d_in + base) = direct(base);
      if (2 * ox + 1 < W) *reinterpret_cast<uint4*>(d_in + base + C) = direct(base + C);
      if (2 * oy + 1 < H) {
        *reinterpret_cast<uint4*>(d_in + base + (int64_t)W * C) = direct(base + (int64_t)W * C);
        if (2 * ox + 1 < W) *reinterpret_cast<uint4*>(d_in + base + (int64_t)W * C + C) = direct(base + (int64_t)W * C + C);
      }
      continue;
    }
    float v[4][8], g[8], o[4][8];
    bf8_unpack(*reinterpret_cast<const uint4*>(in + base), v[0]);
    bf8_unpack(*reinterpret_cast<const uint4*>(in + base + C), v[1]);
    bf8_unpack(*reinterpret_cast<const uint4*>(in + base + (int64_t)W * C), v[2]);
    bf8_unpack(*reinterpret_cast<const uint4*>(in + base + (int64_t)W * C + C), v[3]);
    bf8_unpack(*reinterpret_cast<const uint4*>(d_out + ((int64_t)(b * Ho + oy) * Wo + ox) * C + c8 * 8), g);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int am = 0;
      float m = v[0][k];
#pragma unroll
      for (int j = 1; j < 4; ++j)
        if (v[j][k] > m) {
          m = v[j][k];
          am = j;
        }
      const float gv = g[k] * act_grad_from_out(m, act, alpha);   // `in` is the activation OUTPUT
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j][k] = am == j ? gv : 0.f;
    }
    if (add) {
      const int64_t offs[4] = {base, base + C, base + (int64_t)W * C, base + (int64_t)W * C + C};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a[8];
        bf8_unpack(*reinterpret_cast<const uint4*>(add + offs[j]), a);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[j][k] += a[k] * act_grad_from_out(v[j][k], act, alpha);
      }
    }
    *reinterpret_cast<uint4*>(d_in + base) = bf8_pack(o[0]);
    *reinterpret_cast<uint4*>(d_in + base + C) = bf8_pack(o[1]);
    *reinterpret_cast<uint4*>(d_in + base + (int64_t)W * C) = bf8_pack(o[2]);
    *reinterpret_cast<uint4*>(d_in + base + (int64_t)W * C + C) = bf8_pack(o[3]);
  }
}

// ------------------------------------------------------------------------------------------------
// tf.image.resize_images x2, legacy bilinear -- reference lib/frvsr.py:21-22  [TF1] A.4
template <typename T>
__global__ __launch_bounds__(256) void upsample2_fwd_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H,
                                                            int W, int C) {
  const int Ho = 2 * H, Wo = 2 * W;
  const int64_t n = (int64_t)N * Ho * Wo * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    int64_t t = e / C;
    const int X = (int)(t % Wo);
    t /= Wo;
    const int Y = (int)(t % Ho), b = (int)(t / Ho);
    const int i = Y >> 1, j = X >> 1, i1 = min(i + 1, H - 1), j1 = min(j + 1, W - 1);
    const float ya = 0.5f * (Y & 1), xa = 0.5f * (X & 1);
    const T* __restrict__ base = in + (int64_t)b * H * W * C + c;
    const float tl = Elem<T>::ld(base + ((int64_t)i * W + j) * C), tr = Elem<T>::ld(base + ((int64_t)i * W + j1) * C);
    const float bl = Elem<T>::ld(base + ((int64_t)i1 * W + j) * C), br = Elem<T>::ld(base + ((int64_t)i1 * W + j1) * C);
    const float top = tl + (tr - tl) * xa, bot = bl + (br - bl) * xa;
    Elem<T>::st(out + e, top + (bot - top) * ya);
  }
}

// bf16, C % 8 == 0: one thread per (output pixel, channel octet), 16-byte loads / stores, 32-bit index math; the same
// lerp expressions as the scalar kernel (bit-identical results).
__global__ __launch_bounds__(256) void upsample2_fwd_x8_kernel(const u16* __restrict__ in, u16* __restrict__ out, int N, int H,
                                                               int W, int C) {
  const int Ho = 2 * H, Wo = 2 * W, C8 = C >> 3;
  const int n = N * Ho * Wo * C8;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int c8 = e % C8;
    int t = e / C8;
    const int X = t % Wo;
    t /= Wo;
    const int Y = t % Ho, b = t / Ho;
    const int i = Y >> 1, j = X >> 1, i1 = min(i + 1, H - 1), j1 = min(j + 1, W - 1);
    const float ya = 0.5f * (Y & 1), xa = 0.5f * (X & 1);
    const u16* __restrict__ base = in + (int64_t)b * H * W * C + c8 * 8;
    float tl[8], tr[8], bl[8], br[8], o[8];
    bf8_unpack(*reinterpret_cast<const uint4*>(base + ((int64_t)i * W + j) * C), tl);
    bf8_unpack(*reinterpret_cast<const uint4*>(base + ((int64_t)i * W + j1) * C), tr);
    bf8_unpack(*reinterpret_cast<const uint4*>(base + ((int64_t)i1 * W + j) * C), bl);
    bf8_unpack(*reinterpret_cast<const uint4*>(base + ((int64_t)i1 * W + j1) * C), br);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float top = tl[k] + (tr[k] - tl[k]) * xa, bot = bl[k] + (br[k] - bl[k]) * xa;
      o[k] = top + (bot - top) * ya;
    }
    *reinterpret_cast<uint4*>(out + (int64_t)e * 8) = bf8_pack(o);
  }
}

// contributions of output index o to input index i along one axis: list (o, weight)
__device__ __forceinline__ int up2_terms(int i, int n, int* o, float* wgt) {
  int k = 0;
  o[k] = 2 * i;
  wgt[k++] = 1.f;
  o[k] = 2 * i + 1;
  wgt[k++] = (i == n - 1) ? 1.f : 0.5f;
  if (i >= 1) {
    o[k] = 2 * i - 1;
    wgt[k++] = 0.5f;
  }
  return k;
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const T* __restrict__ d_out, T* __restrict__ d_in, int N,
                                                            int H, int W, int C, const T* __restrict__ y, int act,
                                                            float alpha) {
  const int Ho = 2 * H, Wo = 2 * W;
  const int64_t n = (int64_t)N * H * W * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    int64_t t = e / C;
    const int j = (int)(t % W);
    t /= W;
    const int i = (int)(t % H), b = (int)(t / H);
    int oy[3], ox[3];
    float wy[3], wx[3];
    const int ny = up2_terms(i, H, oy, wy), nx = up2_terms(j, W, ox, wx);
    const T* __restrict__ g = d_out + (int64_t)b * Ho * Wo * C + c;
    float s = 0.f;
    for (int a = 0; a < ny; ++a)
      for (int d = 0; d < nx; ++d) s += wy[a] * wx[d] * Elem<T>::ld(g + ((int64_t)oy[a] * Wo + ox[d]) * C);
    if (y) s *= act_grad_from_out(Elem<T>::ld(y + e), act, alpha);
    Elem<T>::st(d_in + e, s);
  }
}

// ------------------------------------------------------------------------------------------------
// out = (conv_out + bicubic_four(lr)) * 2 - 1 -- reference lib/frvsr.py:81-87, lib/ops.py:166-212
// Keys a=-0.75 weights for t in {0,.25,.5,.75}; all exactly representable (multiples of 2^-8).
__constant__ float kBicubic[4][4] = {{0.f, 1.f, 0.f, 0.f},
                                     {-0.10546875f, 0.87890625f, 0.26171875f, -0.03515625f},
                                     {-0.09375f, 0.59375f, 0.59375f, -0.09375f},
                                     {-0.03515625f, 0.26171875f, 0.87890625f, -0.10546875f}};

template <typename TI>
__global__ __launch_bounds__(256) void bicubic_add_kernel(const float* __restrict__ conv_out,
                                                          const TI* __restrict__ gen_in, int Cpad,
                                                          float* __restrict__ out, int B, int h, int w) {
  const int H = 4 * h, W = 4 * w;
  const int64_t n = (int64_t)B * H * W;
  for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < n; pix += (int64_t)gridDim.x * blockDim.x) {
    const int X = (int)(pix % W), Y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    const int i = Y >> 2, j = X >> 2;
    const float* wy = kBicubic[Y & 3];
    const float* wx = kBicubic[X & 3];
    int ry[4], rx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ry[k] = min(max(i + k - 1, 0), h - 1);   // replicate pad: 1 top/left, 2 bottom/right
      rx[k] = min(max(j + k - 1, 0), w - 1);
    }
    const TI* __restrict__ base = gen_in + (int64_t)b * h * w * Cpad;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float col[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // rows first (lib/ops.py:190-198)
        const float p0 = Elem<TI>::ld(base + ((int64_t)ry[0] * w + rx[k]) * Cpad + c);
        const float p1 = Elem<TI>::ld(base + ((int64_t)ry[1] * w + rx[k]) * Cpad + c);
        const float p2 = Elem<TI>::ld(base + ((int64_t)ry[2] * w + rx[k]) * Cpad + c);
        const float p3 = Elem<TI>::ld(base + ((int64_t)ry[3] * w + rx[k]) * Cpad + c);
        col[k] = wy[0] * p0 + wy[1] * p1 + wy[2] * p2 + wy[3] * p3;
      }
      const float bic = wx[0] * col[0] + wx[1] * col[1] + wx[2] * col[2] + wx[3] * col[3];
      out[pix * 3 + c] = (conv_out[pix * 3 + c] + bic) * 2.f - 1.f;
    }
  }
}

// Row-quad form: one thread per (HR row Y, LR column j) = four horizontally adjacent HR pixels.  They share the 4x4 LR
// neighbourhood (fetched once: 16 pixels instead of 64) and the four row-interpolated columns (the weights of the row pass
// depend on Y & 3 only); conv_out arrives and the frame leaves as 3 x 16-byte vectors.  Optionally also writes the
// deprocessed frame (x + 1) / 2 -- the recurrent state of the inference loop (main.py:207) -- so that pass disappears.
// Same operation order as the per-pixel kernel: rows first, then columns (lib/ops.py:190-210).
// the three colour channels of one LR pixel with ONE load (the row stride is Cpad elements: a scalar load per channel made
// every lane touch its own cache line three times -- the kernel was bound by line requests, 62 us at 1080p)
template <bool VEC>
__device__ __forceinline__ void ld_rgb(const u16* __restrict__ p, float (&v)[3]) {
  if constexpr (VEC) {
    const uint2 q = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(q.x << 16);
    v[1] = __uint_as_float(q.x & 0xffff0000u);
    v[2] = __uint_as_float(q.y << 16);
  } else {
    v[0] = bf2f(p[0]); v[1] = bf2f(p[1]); v[2] = bf2f(p[2]);
  }
}
template <bool VEC>
__device__ __forceinline__ void ld_rgb(const float* __restrict__ p, float (&v)[3]) {
  if constexpr (VEC) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z;
  } else {
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
  }
}

// VEC is a TEMPLATE parameter on purpose: as a run-time flag it put a branch around each of the 16 loads and hipcc waited
// vmcnt(0) per load -- 16 dependent round trips, 95 us instead of 62 (the trap of cdna_hip_programming.md section 5, item 4c).
template <typename TI, bool VEC>
__global__ __launch_bounds__(256) void bicubic_add_quad_kernel(const float* __restrict__ conv_out,
                                                               const TI* __restrict__ gen_in, int Cpad,
                                                               float* __restrict__ out, float* __restrict__ state, int B,
                                                               int h, int w) {
  const int H = 4 * h;
  const int n = B * H * w;                                  // < 2^31 (checked by the host)
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int j = e % w, Y = (e / w) % H, b = e / (w * H);
    const int i = Y >> 2;
    const float* wy = kBicubic[Y & 3];
    int ry[4], rx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ry[k] = min(max(i + k - 1, 0), h - 1);   // replicate pad: 1 top/left, 2 bottom/right
      rx[k] = min(max(j + k - 1, 0), w - 1);
    }
    const TI* __restrict__ base = gen_in + (int64_t)b * h * w * Cpad;
    float col[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float p[4][3];
#pragma unroll
      for (int m = 0; m < 4; ++m) ld_rgb<VEC>(base + ((int64_t)ry[m] * w + rx[k]) * Cpad, p[m]);
#pragma unroll
      for (int c = 0; c < 3; ++c) col[k][c] = wy[0] * p[0][c] + wy[1] * p[1][c] + wy[2] * p[2][c] + wy[3] * p[3][c];
    }
    const int64_t o = ((int64_t)(b * H + Y) * (4 * w) + 4 * j) * 3;          // 12 consecutive floats, 48-byte aligned
    float v[12];
    *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(conv_out + o);
    *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(conv_out + o + 4);
    *reinterpret_cast<float4*>(v + 8) = *reinterpret_cast<const float4*>(conv_out + o + 8);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const float* wx = kBicubic[x];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float bic = wx[0] * col[0][c] + wx[1] * col[1][c] + wx[2] * col[2][c] + wx[3] * col[3][c];
        v[x * 3 + c] = (v[x * 3 + c] + bic) * 2.f - 1.f;
      }
    }
    if (out) {
      *reinterpret_cast<float4*>(out + o) = *reinterpret_cast<float4*>(v);
      *reinterpret_cast<float4*>(out + o + 4) = *reinterpret_cast<float4*>(v + 4);
      *reinterpret_cast<float4*>(out + o + 8) = *reinterpret_cast<float4*>(v + 8);
    }
    if (state) {
#pragma unroll
      for (int k = 0; k < 12; ++k) v[k] = v[k] * 0.5f + 0.5f;
      *reinterpret_cast<float4*>(state + o) = *reinterpret_cast<float4*>(v);
      *reinterpret_cast<float4*>(state + o + 4) = *reinterpret_cast<float4*>(v + 4);
      *reinterpret_cast<float4*>(state + o + 8) = *reinterpret_cast<float4*>(v + 8);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// d_in = scale * d_out * act'(y)   (y nullable -> plain scale + cast)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void act_bwd_kernel(const TI* __restrict__ d_out, const TI* __restrict__ y,
                                                      TO* __restrict__ d_in, int64_t n, int act, float alpha,
                                                      float scale) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    float g = Elem<TI>::ld(d_out + e) * scale;
    if (y) g *= act_grad_from_out(Elem<TI>::ld(y + e), act, alpha);
    Elem<TO>::st(d_in + e, g);
  }
}

__global__ __launch_bounds__(256) void act_bwd_x8_kernel(const u16* __restrict__ d_out, const u16* __restrict__ y,
                                                         u16* __restrict__ d_in, int64_t n8, int act, float alpha,
                                                         float scale) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
    float g[8], yv[8];
    bf8_unpack(*reinterpret_cast<const uint4*>(d_out + e * 8), g);
    if (y) bf8_unpack(*reinterpret_cast<const uint4*>(y + e * 8), yv);
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = g[k] * scale * (y ? act_grad_from_out(yv[k], act, alpha) : 1.f);
    *reinterpret_cast<uint4*>(d_in + e * 8) = bf8_pack(g);
  }
}

// out[pix][0:Ca] = a, [Ca:Ca+Cb] = b, rest 0  (builds the zero-padded NHWC inputs of the first convs)
template <typename TO>
__global__ __launch_bounds__(256) void concat2_pad_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b,
                                                          int Cb, TO* __restrict__ out, int Cpad, int64_t npix,
                                                          float scale) {
  const int64_t n = npix * Cpad;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % Cpad);
    const int64_t pix = e / Cpad;
    float v = 0.f;
    if (c < Ca) v = a[pix * Ca + c];
    else if (c < Ca + Cb) v = b[pix * Cb + c - Ca];
    Elem<TO>::st(out + e, v * scale);
  }
}

__global__ __launch_bounds__(256) void lincomb_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, int64_t n, float alpha, float beta,
                                                      int accumulate) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float v = alpha * a[e] + (b ? beta * b[e] : 0.f);
    out[e] = accumulate ? out[e] + v : v;
  }
}

// ------------------------------------------------------------------------------------------------
// slim.batch_norm(is_training=True, scale=False, eps=1e-3, decay=.9) + lrelu(0.2)
// reference lib/ops.py:88-90, lib/Teco.py:38-39  [TF1] A.7.   x viewed as [rows][C].
// pass 0: mean; pass 1: biased variance around that mean (two-pass, no E[x^2]-E[x]^2 cancellation).
// Per-thread and per-workgroup partial sums in DOUBLE (this generic form is the fp32 parity mode's): the discriminator's
// gradients subtract sums over 1e5 pixels that nearly cancel, and with float partials the D input-conv gradient moved
// between 2e-3 and 5e-3 (relative L2 against the fp64 oracle) from run to run with the order of the atomics.
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, int64_t rows, int C,
                                                       float* __restrict__ stats, int pass) {
  __shared__ double red[256];
  const int Cb = C < 256 ? C : 256, lpc = 256 / Cb;
  const double inv = 1.0 / (double)rows;
  for (int c0 = blockIdx.y * Cb; c0 < C; c0 += gridDim.y * Cb) {
    const int c = c0 + (threadIdx.x % Cb), rsub = threadIdx.x / Cb;
    double s = 0.0;
    if (rsub < lpc && c < C) {
      const float mu = pass ? stats[c] : 0.f;
      for (int64_t r = (int64_t)blockIdx.x * lpc + rsub; r < rows; r += (int64_t)gridDim.x * lpc) {
        const float v = Elem<T>::ld(x + r * C + c) - mu;
        s += pass ? (double)v * v : (double)v;
      }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < Cb && c < C) {
      double t = 0.0;
      for (int k = 0; k < lpc; ++k) t += red[k * Cb + threadIdx.x];
      unsafeAtomicAdd(stats + pass * C + c, (float)(t * inv));
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_lrelu_apply_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows,
                                                             int C, const float* __restrict__ beta,
                                                             const float* __restrict__ stats, float eps, float alpha,
                                                             float* __restrict__ moving) {
  const int64_t n = rows * C;
  if (moving && blockIdx.x == 0) {   // [TF1] moving stats: decay .9, unbiased variance fed to the average
    const float corr = rows > 1 ? (float)rows / (float)(rows - 1) : 1.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      moving[c] = moving[c] * 0.9f + stats[c] * 0.1f;
      moving[C + c] = moving[C + c] * 0.9f + stats[C + c] * corr * 0.1f;
    }
  }
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const float v = (Elem<T>::ld(x + e) - stats[c]) * rsqrtf(stats[C + c] + eps) + beta[c];
    Elem<T>::st(y + e, v > 0.f ? v : v * alpha);
  }
}

// sums[0][c] = sum dz, sums[1][c] = sum dz*xhat  with dz = dy * lrelu'(y)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                          const T* __restrict__ dy, int64_t rows, int C,
                                                          const float* __restrict__ stats, float eps, float alpha,
                                                          float* __restrict__ sums) {
  __shared__ double red0[256], red1[256];                 // double partials: see bn_stats_kernel
  const int Cb = C < 256 ? C : 256, lpc = 256 / Cb;
  for (int c0 = blockIdx.y * Cb; c0 < C; c0 += gridDim.y * Cb) {
    const int c = c0 + (threadIdx.x % Cb), rsub = threadIdx.x / Cb;
    double s0 = 0.0, s1 = 0.0;
    if (rsub < lpc && c < C) {
      const float mu = stats[c], rstd = rsqrtf(stats[C + c] + eps);
      for (int64_t r = (int64_t)blockIdx.x * lpc + rsub; r < rows; r += (int64_t)gridDim.x * lpc) {
        const float dz = Elem<T>::ld(dy + r * C + c) * (Elem<T>::ld(y + r * C + c) > 0.f ? 1.f : alpha);
        s0 += (double)dz;
        s1 += (double)dz * (double)((Elem<T>::ld(x + r * C + c) - mu) * rstd);
      }
    }
    red0[threadIdx.x] = s0;
    red1[threadIdx.x] = s1;
    __syncthreads();
    if (threadIdx.x < Cb && c < C) {
      double t0 = 0.0, t1 = 0.0;
      for (int k = 0; k < lpc; ++k) {
        t0 += red0[k * Cb + threadIdx.x];
        t1 += red1[k * Cb + threadIdx.x];
      }
      unsafeAtomicAdd(sums + c, (float)t0);
      unsafeAtomicAdd(sums + C + c, (float)t1);
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                           const T* __restrict__ dy, T* __restrict__ dx, int64_t rows,
                                                           int C, const float* __restrict__ stats, float eps,
                                                           float alpha, const float* __restrict__ sums,
                                                           float* __restrict__ d_beta) {
  const int64_t n = rows * C;
  const float inv = 1.f / (float)rows;
  if (d_beta && blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += blockDim.x) d_beta[c] += sums[c];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const float rstd = rsqrtf(stats[C + c] + eps);
    const float xhat = (Elem<T>::ld(x + e) - stats[c]) * rstd;
    const float dz = Elem<T>::ld(dy + e) * (Elem<T>::ld(y + e) > 0.f ? 1.f : alpha);
    Elem<T>::st(dx + e, rstd * (dz - sums[c] * inv - xhat * sums[C + c] * inv));
  }
}

// ---- bf16, C % 8 == 0 with C/8 a power of two <= 32 (the D widths 64/128/256): 16-byte accesses, every thread owns one
//      channel octet for the whole launch (its per-channel constants live in registers), <= 256 workgroups per
//      reduction so that the per-channel atomics of a launch stay few (same-address atomics serialise).
template <int PASS>   // 0: mean, 1: biased variance around stats[c]
__global__ __launch_bounds__(256) void bn_stats_x8_kernel(const u16* __restrict__ x, int64_t rows, int C,
                                                          float* __restrict__ stats) {
  __shared__ float red[256 * 8];
  const int OC = C >> 3, RP = 256 / OC, oc = threadIdx.x % OC, rsub = threadIdx.x / OC;
  float mu[8], s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    mu[k] = PASS ? stats[oc * 8 + k] : 0.f;
    s[k] = 0.f;
  }
  for (int64_t r = (int64_t)blockIdx.x * RP + rsub; r < rows; r += (int64_t)gridDim.x * RP) {
    float v[8];
    bf8_unpack(*reinterpret_cast<const uint4*>(x + r * C + oc * 8), v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float d = v[k] - mu[k];
      s[k] += PASS ? d * d : d;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = s[k];
  __syncthreads();
  const float inv = 1.f / (float)rows;
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int k = 0; k < RP; ++k) t += red[(k * OC + (c >> 3)) * 8 + (c & 7)];
    unsafeAtomicAdd(stats + PASS * C + c, t * inv);
  }
}

// stats[r][0][c] / stats[r][1][c], r < TG_BN_STAT_REPLICAS: partial E[x] / E[x^2] accumulated by the producing conv's epilogue
// -> stats[0] = [mean, biased variance E[x^2] - mean^2]
// Deliberate deviation from tf.nn.moments (two-pass, lib/ops.py:89 through slim.batch_norm): single-pass moments of the conv's fp32
// accumulators, clamped at 0.  The cancellation error is eps_fp32 * mean^2 absolute; held by
// tests/test_kernels_gpu.py::test_conv4x4s2_fused_bn_statistics_with_large_mean_small_variance_channels at |mean| / std = 80 (1 %).
__global__ __launch_bounds__(256) void bn_moment_to_var_kernel(float* __restrict__ stats, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float m = 0.f, q = 0.f;
#pragma unroll
  for (int r = 0; r < TG_BN_STAT_REPLICAS; ++r) {
    m += stats[(size_t)r * 2 * C + c];
    q += stats[(size_t)r * 2 * C + C + c];
  }
  stats[c] = m;
  stats[C + c] = fmaxf(q - m * m, 0.f);
}

__global__ __launch_bounds__(256) void bn_lrelu_apply_x8_kernel(const u16* __restrict__ x, u16* __restrict__ y,
                                                                int64_t rows, int C, const float* __restrict__ beta,
                                                                const float* __restrict__ stats, float eps, float alpha,
                                                                float* __restrict__ moving) {
  if (moving && blockIdx.x == 0) {   // [TF1] moving stats: decay .9, unbiased variance fed to the average
    const float corr = rows > 1 ? (float)rows / (float)(rows - 1) : 1.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      moving[c] = moving[c] * 0.9f + stats[c] * 0.1f;
      moving[C + c] = moving[C + c] * 0.9f + stats[C + c] * corr * 0.1f;
    }
  }
  const int OC = C >> 3;
  const int64_t n8 = rows * OC;
  const int oc = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % OC);     // constant per thread: 256 % OC == 0
  float mu[8], rs[8], bt[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    mu[k] = stats[oc * 8 + k];
    rs[k] = rsqrtf(stats[C + oc * 8 + k] + eps);
    bt[k] = beta[oc * 8 + k];
  }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
    float v[8];
    bf8_unpack(*reinterpret_cast<const uint4*>(x + e * 8), v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float t = (v[k] - mu[k]) * rs[k] + bt[k];
      v[k] = t > 0.f ? t : t * alpha;
    }
    *reinterpret_cast<uint4*>(y + e * 8) = bf8_pack(v);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_sums_x8_kernel(const u16* __restrict__ x, const u16* __restrict__ y,
                                                             const u16* __restrict__ dy, int64_t rows, int C,
                                                             const float* __restrict__ stats, float eps, float alpha,
                                                             float* __restrict__ sums) {
  __shared__ float red0[256 * 8], red1[256 * 8];
  const int OC = C >> 3, RP = 256 / OC, oc = threadIdx.x % OC, rsub = threadIdx.x / OC;
  float mu[8], rs[8], s0[8], s1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    mu[k] = stats[oc * 8 + k];
    rs[k] = rsqrtf(stats[C + oc * 8 + k] + eps);
    s0[k] = s1[k] = 0.f;
  }
  for (int64_t r = (int64_t)blockIdx.x * RP + rsub; r < rows; r += (int64_t)gridDim.x * RP) {
    float xv[8], yv[8], gv[8];
    const int64_t off = r * C + oc * 8;
    bf8_unpack(*reinterpret_cast<const uint4*>(x + off), xv);
    bf8_unpack(*reinterpret_cast<const uint4*>(y + off), yv);
    bf8_unpack(*reinterpret_cast<const uint4*>(dy + off), gv);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float dz = gv[k] * (yv[k] > 0.f ? 1.f : alpha);
      s0[k] += dz;
      s1[k] += dz * (xv[k] - mu[k]) * rs[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    red0[threadIdx.x * 8 + k] = s0[k];
    red1[threadIdx.x * 8 + k] = s1[k];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t0 = 0.f, t1 = 0.f;
    for (int k = 0; k < RP; ++k) {
      t0 += red0[(k * OC + (c >> 3)) * 8 + (c & 7)];
      t1 += red1[(k * OC + (c >> 3)) * 8 + (c & 7)];
    }
    unsafeAtomicAdd(sums + c, t0);
    unsafeAtomicAdd(sums + C + c, t1);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_x8_kernel(const u16* __restrict__ x, const u16* __restrict__ y,
                                                              const u16* __restrict__ dy, u16* __restrict__ dx,
                                                              int64_t rows, int C, const float* __restrict__ stats,
                                                              float eps, float alpha, const float* __restrict__ sums,
                                                              float* __restrict__ d_beta) {
  const float inv = 1.f / (float)rows;
  if (d_beta && blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += blockDim.x) d_beta[c] += sums[c];
  const int OC = C >> 3;
  const int64_t n8 = rows * OC;
  const int oc = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % OC);
  float mu[8], rs[8], m0[8], m1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    mu[k] = stats[oc * 8 + k];
    rs[k] = rsqrtf(stats[C + oc * 8 + k] + eps);
    m0[k] = sums[oc * 8 + k] * inv;
    m1[k] = sums[C + oc * 8 + k] * inv;
  }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
    float xv[8], yv[8], gv[8];
    bf8_unpack(*reinterpret_cast<const uint4*>(x + e * 8), xv);
    bf8_unpack(*reinterpret_cast<const uint4*>(y + e * 8), yv);
    bf8_unpack(*reinterpret_cast<const uint4*>(dy + e * 8), gv);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float xhat = (xv[k] - mu[k]) * rs[k];
      const float dz = gv[k] * (yv[k] > 0.f ? 1.f : alpha);
      gv[k] = rs[k] * (dz - m0[k] - xhat * m1[k]);
    }
    *reinterpret_cast<uint4*>(dx + e * 8) = bf8_pack(gv);
  }
}

// ------------------------------------------------------------------------------------------------
// tf.train.AdamOptimizer (reference lib/Teco.py:425,439-440)  [TF1] A.11
// hyper = {lr_t, beta1, beta2, eps, gate}; gate == 0 leaves p, m, v untouched (tf.cond D-gate).
__global__ __launch_bounds__(256) void adam_tf_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                      const float* __restrict__ hyper, float grad_scale) {
  const float lr_t = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3];
  if (hyper[4] == 0.f) return;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float gr = g[e] * grad_scale;
    const float mm = m[e] * b1 + gr * (1.f - b1);
    const float vv = v[e] * b2 + gr * gr * (1.f - b2);
    m[e] = mm;
    v[e] = vv;
    p[e] -= lr_t * mm / (sqrtf(vv) + eps);
  }
}

// ------------------------------------------------------------------------------------------------
// weight re-layout: dst[tap][b][a] = src[tap][a][b] (or plain converting copy); table-driven so one
// launch re-packs every weight of the flat parameter buffer.
template <typename TD>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ src, TD* __restrict__ dstT, TD* __restrict__ dstN,
                                                           const int64_t* __restrict__ tab, int transpose) {
  // blockIdx.z selects the layout when BOTH compute copies are refreshed in one launch (transpose < 0): z = 0 the transposed
  // copy into dstT, z = 1 the natural copy into dstN; otherwise `transpose` says which and dstT is the destination
  const bool tr = transpose < 0 ? blockIdx.z == 0 : transpose != 0;
  TD* __restrict__ dst = (transpose < 0 && blockIdx.z == 1) ? dstN : dstT;
  const int64_t* t = tab + (int64_t)blockIdx.y * 7;
  const int64_t so = t[0], dof = t[1];
  const unsigned taps = (unsigned)t[2];
  const unsigned A = (unsigned)t[3], Bs = (unsigned)t[4], Ap = (unsigned)t[5];   // src [tap][A][B]; A zero-padded to Ap in dst
  const unsigned Bd = tr ? Bs : (unsigned)t[6];                                  // natural copy: B zero-padded to Bp as well
  const unsigned n = taps * Ap * Bd;                          // (a weight tensor has far fewer than 2^32 elements: 32-bit index math)
  const float* __restrict__ s0 = src + so;
  TD* __restrict__ d0 = dst + dof;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    unsigned a, b, tap;
    if (tr) {         // e indexes dst [tap][b][a_pad]
      const unsigned r = e / Ap;
      a = e - r * Ap;
      tap = r / Bd;
      b = r - tap * Bd;
    } else {          // e indexes dst [tap][a_pad][b]
      const unsigned r = e / Bd;
      b = e - r * Bd;
      tap = r / Ap;
      a = r - tap * Ap;
    }
    Elem<TD>::st(d0 + e, (a < A && b < Bs) ? s0[(tap * A + a) * Bs + b] : 0.f);
  }
}

// Fragment-order copies of 64 -> 64 3x3 weights for csrc/resblock_lat.hip: element [s = 2 tap + kk][wave][lane][j] of a copy is
// W[tap][row = 16 wave + lane % 16][k = 32 kk + 8 (lane / 16) + j] -- exactly the 16 bytes lane `lane` of wave `wave` feeds to
// the MFMA of step s, so that a wave's weight load is ONE contiguous KiB.  z = 0: the forward operand (row = out channel,
// k = in channel), z = 1: the input-gradient operand (row = in channel, k = out channel); src is TF's HWIO fp32 master copy.
__global__ __launch_bounds__(256) void pack_weights_frag_kernel(const float* __restrict__ src, u16* __restrict__ dstT,
                                                                u16* __restrict__ dstN, const int64_t* __restrict__ tab) {
  const int64_t* t = tab + (int64_t)blockIdx.y * 3;
  const float* __restrict__ s0 = src + t[0];
  const unsigned cin = (unsigned)t[2];                      // HWIO [3,3,cin,64]; channels cin .. 63 of the copies are zero
  u16* __restrict__ d0 = (blockIdx.z == 0 ? dstT : dstN) + t[1];
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < 9u * 64u * 64u; e += gridDim.x * blockDim.x) {
    const unsigned j = e & 7, lane = (e >> 3) & 63, wave = (e >> 9) & 3, s = e >> 11;
    const unsigned tap = s >> 1, kk = s & 1;
    const unsigned r = wave * 16 + (lane & 15), k = kk * 32 + (lane >> 4) * 8 + j;
    const unsigned ci = blockIdx.z == 0 ? k : r, co = blockIdx.z == 0 ? r : k;
    const float v = ci < cin ? s0[(tap * cin + ci) * 64 + co] : 0.f;
    d0[e] = f2bf(v);
  }
}

// ------------------------------------------------------------------------------------------------
// loss reductions: out[0] += scale * sum f(a-b)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void sum_diff_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t n,
                                                       float scale, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float d = Elem<T>::ld(a + e) - Elem<T>::ld(b + e);
    s += MODE == 0 ? d * d : fabsf(d);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, (red[0] + red[1] + red[2] + red[3]) * scale);
}

// ================================================================================================
#define ST(s) static_cast<hipStream_t>(s)
#define DISPATCH_DT(dtype, KERNEL, grid, ...)                                                    \
  if ((dtype) == TG_F32) hipLaunchKernelGGL((KERNEL<float>), grid, dim3(256), 0, ST(stream), __VA_ARGS__); \
  else if ((dtype) == TG_BF16) hipLaunchKernelGGL((KERNEL<u16>), grid, dim3(256), 0, ST(stream), __VA_ARGS__); \
  else TG_CHECK_ARG(false, "bad dtype")

extern "C" int tg_maxpool2_forward(const void* in, void* out, int dtype, int N, int H, int W, int C, void* stream) {
  TG_CHECK_ARG(in && out && N > 0 && H > 1 && W > 1 && C > 0, "bad argument");
  const bool x8 = dtype == TG_BF16 && C % 8 == 0 && ((((uintptr_t)in | (uintptr_t)out)) & 15) == 0 &&
                  (int64_t)N * H * W * C < ((int64_t)1 << 31);
  if (x8) {
    hipLaunchKernelGGL(maxpool2_fwd_x8_kernel, dim3(grid_1d((int64_t)N * (H / 2) * (W / 2) * (C / 8), 256, 1 << 20)), dim3(256),
                       0, ST(stream), (const u16*)in, (u16*)out, N, H, W, C);
    TG_CHECK_LAUNCH();
  }
  dim3 grid(grid_1d((int64_t)N * (H / 2) * (W / 2) * C, 256));
  if (dtype == TG_F32) hipLaunchKernelGGL((maxpool2_fwd_kernel<float>), grid, dim3(256), 0, ST(stream), (const float*)in, (float*)out, N, H, W, C);
  else if (dtype == TG_BF16) hipLaunchKernelGGL((maxpool2_fwd_kernel<u16>), grid, dim3(256), 0, ST(stream), (const u16*)in, (u16*)out, N, H, W, C);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

extern "C" int tg_maxpool2_backward(const void* in, const void* d_out, void* d_in, int dtype, int N, int H, int W,
                                    int C, int act, float alpha, const void* add, void* stream) {
  TG_CHECK_ARG(in && d_out && d_in && N > 0 && H > 1 && W > 1 && C > 0, "bad argument");
  const bool x8 = dtype == TG_BF16 && C % 8 == 0 &&
                  ((((uintptr_t)in | (uintptr_t)d_out | (uintptr_t)d_in | (uintptr_t)add)) & 15) == 0 &&
                  (int64_t)N * H * W * C < ((int64_t)1 << 31);
  if (x8) {
    hipLaunchKernelGGL(maxpool2_bwd_x8_kernel,
                       dim3(grid_1d((int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8), 256, 1 << 20)), dim3(256), 0,
                       ST(stream), (const u16*)in, (const u16*)d_out, (u16*)d_in, N, H, W, C, act, alpha, (const u16*)add);
    TG_CHECK_LAUNCH();
  }
  dim3 grid(grid_1d((int64_t)N * H * W * C, 256));
  if (dtype == TG_F32) hipLaunchKernelGGL((maxpool2_bwd_kernel<float>), grid, dim3(256), 0, ST(stream), (const float*)in, (const float*)d_out, (float*)d_in, N, H, W, C, act, alpha, (const float*)add);
  else if (dtype == TG_BF16) hipLaunchKernelGGL((maxpool2_bwd_kernel<u16>), grid, dim3(256), 0, ST(stream), (const u16*)in, (const u16*)d_out, (u16*)d_in, N, H, W, C, act, alpha, (const u16*)add);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

extern "C" int tg_upsample2_forward(const void* in, void* out, int dtype, int N, int H, int W, int C, void* stream) {
  TG_CHECK_ARG(in && out && N > 0 && H > 0 && W > 0 && C > 0, "bad argument");
  dim3 grid(grid_1d((int64_t)N * H * W * 4 * C, 256));
  const double by = (double)N * H * W * C * 5.0 * (dtype == TG_F32 ? 4.0 : 2.0);
  if (dtype == TG_BF16 && C % 8 == 0 && ((((uintptr_t)in | (uintptr_t)out)) & 15) == 0 &&
      (int64_t)N * H * W * C * 4 < ((int64_t)1 << 31)) {
    TG_LAUNCH("upsample2_fwd_x8", 0, by, upsample2_fwd_x8_kernel, dim3(grid_1d((int64_t)N * H * W * 4 * (C / 8), 256, 1 << 20)),
              dim3(256), 0, ST(stream), (const u16*)in, (u16*)out, N, H, W, C);
    TG_CHECK_LAUNCH();
  }
  if (dtype == TG_F32) TG_LAUNCH("upsample2_fwd<f32>", 0, by, (upsample2_fwd_kernel<float>), grid, dim3(256), 0, ST(stream), (const float*)in, (float*)out, N, H, W, C);
  else if (dtype == TG_BF16) TG_LAUNCH("upsample2_fwd<bf16>", 0, by, (upsample2_fwd_kernel<u16>), grid, dim3(256), 0, ST(stream), (const u16*)in, (u16*)out, N, H, W, C);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

extern "C" int tg_upsample2_backward(const void* d_out, void* d_in, int dtype, int N, int H, int W, int C,
                                     const void* y, int act, float alpha, void* stream) {
  TG_CHECK_ARG(d_out && d_in && N > 0 && H > 0 && W > 0 && C > 0, "bad argument");
  dim3 grid(grid_1d((int64_t)N * H * W * C, 256));
  if (dtype == TG_F32) hipLaunchKernelGGL((upsample2_bwd_kernel<float>), grid, dim3(256), 0, ST(stream), (const float*)d_out, (float*)d_in, N, H, W, C, (const float*)y, act, alpha);
  else if (dtype == TG_BF16) hipLaunchKernelGGL((upsample2_bwd_kernel<u16>), grid, dim3(256), 0, ST(stream), (const u16*)d_out, (u16*)d_in, N, H, W, C, (const u16*)y, act, alpha);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

extern "C" int tg_bicubic_add_preprocess(const float* conv_out, const void* gen_in, int in_dtype, int Cpad, float* out,
                                         float* state, int B, int h, int w, void* stream) {
  TG_CHECK_ARG(conv_out && gen_in && (out || state) && B > 0 && h > 0 && w > 0 && Cpad >= 3, "bad argument");
  dim3 grid(grid_1d((int64_t)B * h * w * 16, 256));
  const double by = (double)B * h * w * (16.0 * 12.0 * (1 + (out != nullptr) + (state != nullptr)) +
                                         3.0 * (in_dtype == TG_F32 ? 4.0 : 2.0));   // conv_out in, frame / state out, LR in
  const bool no_quad = false;
  if (!no_quad && ((((uintptr_t)conv_out | (uintptr_t)out | (uintptr_t)state)) & 15) == 0 &&
      (int64_t)B * h * w * 4 < ((int64_t)1 << 31)) {
    dim3 gq(grid_1d((int64_t)B * h * w * 4, 256, 1 << 20));
    const bool vec = Cpad >= 4 && Cpad % 4 == 0 && (((uintptr_t)gen_in) & 15) == 0;      // one 8 / 16-byte load per LR pixel
    if (in_dtype == TG_F32 && vec) TG_LAUNCH("bicubic_add_quad<f32>", 0, by, (bicubic_add_quad_kernel<float, true>), gq, dim3(256), 0, ST(stream), conv_out, (const float*)gen_in, Cpad, out, state, B, h, w);
    else if (in_dtype == TG_F32) TG_LAUNCH("bicubic_add_quad<f32>", 0, by, (bicubic_add_quad_kernel<float, false>), gq, dim3(256), 0, ST(stream), conv_out, (const float*)gen_in, Cpad, out, state, B, h, w);
    else if (in_dtype == TG_BF16 && vec) TG_LAUNCH("bicubic_add_quad<bf16>", 0, by, (bicubic_add_quad_kernel<u16, true>), gq, dim3(256), 0, ST(stream), conv_out, (const u16*)gen_in, Cpad, out, state, B, h, w);
    else if (in_dtype == TG_BF16) TG_LAUNCH("bicubic_add_quad<bf16>", 0, by, (bicubic_add_quad_kernel<u16, false>), gq, dim3(256), 0, ST(stream), conv_out, (const u16*)gen_in, Cpad, out, state, B, h, w);
    else TG_CHECK_ARG(false, "bad dtype");
    TG_CHECK_LAUNCH();
  }
  TG_CHECK_ARG(out != nullptr && state == nullptr, "the per-pixel fallback writes `out` only");
  if (in_dtype == TG_F32) TG_LAUNCH("bicubic_add<f32>", 0, by, (bicubic_add_kernel<float>), grid, dim3(256), 0, ST(stream), conv_out, (const float*)gen_in, Cpad, out, B, h, w);
  else if (in_dtype == TG_BF16) TG_LAUNCH("bicubic_add<bf16>", 0, by, (bicubic_add_kernel<u16>), grid, dim3(256), 0, ST(stream), conv_out, (const u16*)gen_in, Cpad, out, B, h, w);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

extern "C" int tg_act_backward(const void* d_out, const void* y, void* d_in, int in_dtype, int out_dtype, int64_t n,
                               int act, float alpha, float scale, void* stream) {
  TG_CHECK_ARG(d_out && d_in && n > 0, "bad argument");
  if (in_dtype == TG_BF16 && out_dtype == TG_BF16 && n % 8 == 0 &&
      ((((uintptr_t)d_out | (uintptr_t)y | (uintptr_t)d_in)) & 15) == 0) {
    hipLaunchKernelGGL(act_bwd_x8_kernel, dim3(grid_1d(n / 8, 256, 1 << 20)), dim3(256), 0, ST(stream), (const u16*)d_out,
                       (const u16*)y, (u16*)d_in, n / 8, act, alpha, scale);
    TG_CHECK_LAUNCH();
  }
  dim3 grid(grid_1d(n, 256));
  if (in_dtype == TG_F32 && out_dtype == TG_F32) hipLaunchKernelGGL((act_bwd_kernel<float, float>), grid, dim3(256), 0, ST(stream), (const float*)d_out, (const float*)y, (float*)d_in, n, act, alpha, scale);
  else if (in_dtype == TG_F32 && out_dtype == TG_BF16) hipLaunchKernelGGL((act_bwd_kernel<float, u16>), grid, dim3(256), 0, ST(stream), (const float*)d_out, (const float*)y, (u16*)d_in, n, act, alpha, scale);
  else if (in_dtype == TG_BF16 && out_dtype == TG_BF16) hipLaunchKernelGGL((act_bwd_kernel<u16, u16>), grid, dim3(256), 0, ST(stream), (const u16*)d_out, (const u16*)y, (u16*)d_in, n, act, alpha, scale);
  else if (in_dtype == TG_BF16 && out_dtype == TG_F32) hipLaunchKernelGGL((act_bwd_kernel<u16, float>), grid, dim3(256), 0, ST(stream), (const u16*)d_out, (const u16*)y, (float*)d_in, n, act, alpha, scale);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

extern "C" int tg_concat2_pad(const float* a, int Ca, const float* b, int Cb, void* out, int out_dtype, int Cpad,
                              int64_t npix, float scale, void* stream) {
  TG_CHECK_ARG(a && out && Ca > 0 && Cb >= 0 && (b || Cb == 0) && Cpad >= Ca + Cb && npix > 0, "bad argument");
  dim3 grid(grid_1d(npix * Cpad, 256));
  if (out_dtype == TG_F32) hipLaunchKernelGGL((concat2_pad_kernel<float>), grid, dim3(256), 0, ST(stream), a, Ca, b, Cb, (float*)out, Cpad, npix, scale);
  else if (out_dtype == TG_BF16) hipLaunchKernelGGL((concat2_pad_kernel<u16>), grid, dim3(256), 0, ST(stream), a, Ca, b, Cb, (u16*)out, Cpad, npix, scale);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

// Frame-major gather of a batch-major sequence: dst[t][b][:] = src[b][idx[t]][:] -- the ping-pong extension
// tf.concat(r, r[:, -2::-1]) of reference lib/Teco.py:80-85 plus the [B,T] -> [T,B] re-layout of this path's sequences.
struct SeqIdx { int v[64]; };
template <typename V>      // float4 when a frame is a whole number of 16-byte vectors, float otherwise (odd crop sizes)
__global__ __launch_bounds__(256) void seq_gather_kernel(const V* __restrict__ src, V* __restrict__ dst, int B, int T0,
                                                         int T, int E, SeqIdx idx) {
  const int n = T * B * E;                                    // < 2^31 (checked by the host)
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int k = e % E, tb = e / E;
    const int b = tb % B, t = tb / B;
    dst[e] = src[((int64_t)b * T0 + idx.v[t]) * E + k];
  }
}

extern "C" int tg_seq_gather(const float* src, float* dst, int B, int T0, int T, int64_t frame_elems, const int* idx,
                             void* stream) {
  TG_CHECK_ARG(src && dst && idx && B > 0 && T0 > 0 && T > 0 && T <= 64 && frame_elems > 0, "bad argument");
  const bool vec = frame_elems % 4 == 0 && ((((uintptr_t)src | (uintptr_t)dst)) & 15) == 0;
  const int64_t E64 = vec ? frame_elems / 4 : frame_elems;
  TG_CHECK_ARG((int64_t)T * B * E64 < ((int64_t)1 << 31), "sequence too large");
  SeqIdx s;
  for (int t = 0; t < T; ++t) {
    TG_CHECK_ARG(idx[t] >= 0 && idx[t] < T0, "frame index out of range");
    s.v[t] = idx[t];
  }
  const int E = (int)E64;
  const dim3 grid(grid_1d((int64_t)T * B * E, 256, 8192));
  if (vec)
    hipLaunchKernelGGL(seq_gather_kernel<float4>, grid, dim3(256), 0, ST(stream), (const float4*)src, (float4*)dst, B, T0, T, E, s);
  else
    hipLaunchKernelGGL(seq_gather_kernel<float>, grid, dim3(256), 0, ST(stream), src, dst, B, T0, T, E, s);
  TG_CHECK_LAUNCH();
}

// out = x * scale + shift  (deprocess (x+1)/2 of reference lib/ops.py:19-22 == x*0.5+0.5 bit for bit)
__global__ __launch_bounds__(256) void affine_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n,
                                                     float scale, float shift) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = x[e] * scale + shift;
}

extern "C" int tg_affine(const float* x, float* out, int64_t n, float scale, float shift, void* stream) {
  TG_CHECK_ARG(x && out && n > 0, "bad argument");
  hipLaunchKernelGGL(affine_kernel, dim3(grid_1d(n, 256)), dim3(256), 0, ST(stream), x, out, n, scale, shift);
  TG_CHECK_LAUNCH();
}

extern "C" int tg_lincomb(const float* a, const float* b, float* out, int64_t n, float alpha, float beta,
                          int accumulate, void* stream) {
  TG_CHECK_ARG(a && out && n > 0, "bad argument");
  hipLaunchKernelGGL(lincomb_kernel, dim3(grid_1d(n, 256)), dim3(256), 0, ST(stream), a, b, out, n, alpha, beta, accumulate);
  TG_CHECK_LAUNCH();
}

static bool bn_x8_ok(int dtype, int C, const void* a, const void* b, const void* c, const void* d) {
  const int oc = C / 8;
  return dtype == TG_BF16 && C % 8 == 0 && oc >= 1 && oc <= 32 && (oc & (oc - 1)) == 0 &&
         ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d)) & 15) == 0;
}

static dim3 reduce_grid(int64_t rows, int C) {
  const int Cb = C < 256 ? C : 256, lpc = 256 / Cb;
  int gx = (int)cdiv64(rows, (int64_t)lpc * 32);
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  return dim3(gx, (C + Cb - 1) / Cb);
}

extern "C" int tg_bn_lrelu_forward(const void* x, void* y, int dtype, int64_t rows, int C, const float* beta, float eps,
                                   float alpha, float* stats, float* moving, int prezeroed, void* stream) {
  TG_CHECK_ARG(x && y && beta && stats && rows > 0 && C > 0, "bad argument");
  TG_CHECK_ARG(dtype == TG_F32 || dtype == TG_BF16, "bad dtype");
  if (!prezeroed && hipMemsetAsync(stats, 0, sizeof(float) * 2 * C, ST(stream)) != hipSuccess) {
    tg_set_error("tg_bn_lrelu_forward: memset failed");
    return TG_ELAUNCH;
  }
  TG_CHECK_ARG(prezeroed != 2 || bn_x8_ok(dtype, C, x, y, nullptr, nullptr), "prezeroed = 2 (statistics given): bf16, C % 8 == 0 only");
  if (prezeroed == 2) {
    const int OC = C / 8;
    const dim3 eg8(grid_1d(rows * OC, 256, 2048));
    hipLaunchKernelGGL(bn_moment_to_var_kernel, dim3((C + 255) / 256), dim3(256), 0, ST(stream), stats, C);
    hipLaunchKernelGGL(bn_lrelu_apply_x8_kernel, eg8, dim3(256), 0, ST(stream), (const u16*)x, (u16*)y, rows, C, beta, stats,
                       eps, alpha, moving);
    TG_CHECK_LAUNCH();
  }
  if (bn_x8_ok(dtype, C, x, y, nullptr, nullptr)) {
    const int OC = C / 8, RP = 256 / OC;
    // reductions: RP * 4 rows per workgroup (round 4; it was RP * 16 under a cap of 256 workgroups: D's [12,64,64,64] maps ran
    // as 96 workgroups x 16 dependent trips, 20 us for 19 MB -- latency, not bandwidth)
    const dim3 rg8(grid_1d(rows, RP * 4, 1024)), eg8(grid_1d(rows * OC, 256, 2048));
    hipLaunchKernelGGL((bn_stats_x8_kernel<0>), TG_DET_GRID(rg8), dim3(256), 0, ST(stream), (const u16*)x, rows, C, stats);
    hipLaunchKernelGGL((bn_stats_x8_kernel<1>), TG_DET_GRID(rg8), dim3(256), 0, ST(stream), (const u16*)x, rows, C, stats);
    hipLaunchKernelGGL(bn_lrelu_apply_x8_kernel, eg8, dim3(256), 0, ST(stream), (const u16*)x, (u16*)y, rows, C, beta, stats,
                       eps, alpha, moving);
    TG_CHECK_LAUNCH();
  }
  const dim3 rg = reduce_grid(rows, C);
  const dim3 eg(grid_1d(rows * C, 256));
  if (dtype == TG_F32) {
    hipLaunchKernelGGL((bn_stats_kernel<float>), TG_DET_GRID(rg), dim3(256), 0, ST(stream), (const float*)x, rows, C, stats, 0);
    hipLaunchKernelGGL((bn_stats_kernel<float>), TG_DET_GRID(rg), dim3(256), 0, ST(stream), (const float*)x, rows, C, stats, 1);
    hipLaunchKernelGGL((bn_lrelu_apply_kernel<float>), eg, dim3(256), 0, ST(stream), (const float*)x, (float*)y, rows, C, beta, stats, eps, alpha, moving);
  } else {
    hipLaunchKernelGGL((bn_stats_kernel<u16>), TG_DET_GRID(rg), dim3(256), 0, ST(stream), (const u16*)x, rows, C, stats, 0);
    hipLaunchKernelGGL((bn_stats_kernel<u16>), TG_DET_GRID(rg), dim3(256), 0, ST(stream), (const u16*)x, rows, C, stats, 1);
    hipLaunchKernelGGL((bn_lrelu_apply_kernel<u16>), eg, dim3(256), 0, ST(stream), (const u16*)x, (u16*)y, rows, C, beta, stats, eps, alpha, moving);
  }
  TG_CHECK_LAUNCH();
}

extern "C" int tg_bn_lrelu_backward(const void* x, const void* y, const void* d_y, void* d_x, int dtype, int64_t rows,
                                    int C, const float* stats, float eps, float alpha, float* d_beta, float* ws,
                                    int prezeroed, void* stream) {
  TG_CHECK_ARG(x && y && d_y && d_x && stats && ws && rows > 0 && C > 0, "bad argument");
  TG_CHECK_ARG(dtype == TG_F32 || dtype == TG_BF16, "bad dtype");
  if (!prezeroed && hipMemsetAsync(ws, 0, sizeof(float) * 2 * C, ST(stream)) != hipSuccess) {
    tg_set_error("tg_bn_lrelu_backward: memset failed");
    return TG_ELAUNCH;
  }
  if (bn_x8_ok(dtype, C, x, y, d_y, d_x)) {
    const int OC = C / 8, RP = 256 / OC;
    // reductions: RP * 4 rows per workgroup (round 4; it was RP * 16 under a cap of 256 workgroups: D's [12,64,64,64] maps ran
    // as 96 workgroups x 16 dependent trips, 20 us for 19 MB -- latency, not bandwidth)
    const dim3 rg8(grid_1d(rows, RP * 4, 1024)), eg8(grid_1d(rows * OC, 256, 2048));
    hipLaunchKernelGGL(bn_bwd_sums_x8_kernel, TG_DET_GRID(rg8), dim3(256), 0, ST(stream), (const u16*)x, (const u16*)y, (const u16*)d_y, rows,
                       C, stats, eps, alpha, ws);
    hipLaunchKernelGGL(bn_bwd_apply_x8_kernel, eg8, dim3(256), 0, ST(stream), (const u16*)x, (const u16*)y, (const u16*)d_y,
                       (u16*)d_x, rows, C, stats, eps, alpha, ws, d_beta);
    TG_CHECK_LAUNCH();
  }
  const dim3 rg = reduce_grid(rows, C);
  const dim3 eg(grid_1d(rows * C, 256));
  if (dtype == TG_F32) {
    hipLaunchKernelGGL((bn_bwd_sums_kernel<float>), TG_DET_GRID(rg), dim3(256), 0, ST(stream), (const float*)x, (const float*)y, (const float*)d_y, rows, C, stats, eps, alpha, ws);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), eg, dim3(256), 0, ST(stream), (const float*)x, (const float*)y, (const float*)d_y, (float*)d_x, rows, C, stats, eps, alpha, ws, d_beta);
  } else {
    hipLaunchKernelGGL((bn_bwd_sums_kernel<u16>), TG_DET_GRID(rg), dim3(256), 0, ST(stream), (const u16*)x, (const u16*)y, (const u16*)d_y, rows, C, stats, eps, alpha, ws);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<u16>), eg, dim3(256), 0, ST(stream), (const u16*)x, (const u16*)y, (const u16*)d_y, (u16*)d_x, rows, C, stats, eps, alpha, ws, d_beta);
  }
  TG_CHECK_LAUNCH();
}

extern "C" int tg_adam_tf(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                          float grad_scale, void* stream) {
  TG_CHECK_ARG(p && g && m && v && hyper && n > 0, "bad argument");
  hipLaunchKernelGGL(adam_tf_kernel, dim3(grid_1d(n, 256, 2048)), dim3(256), 0, ST(stream), p, g, m, v, n, hyper,
                     grad_scale);
  TG_CHECK_LAUNCH();
}

extern "C" int tg_pack_weights(const float* src_base, void* dst_base, int dst_dtype, const int64_t* tab, int count,
                               int transpose, void* stream) {
  TG_CHECK_ARG(src_base && dst_base && tab && count > 0, "bad argument");
  dim3 grid(64, count);
  if (dst_dtype == TG_F32) hipLaunchKernelGGL((pack_weights_kernel<float>), grid, dim3(256), 0, ST(stream), src_base, (float*)dst_base, (float*)nullptr, tab, transpose ? 1 : 0);
  else if (dst_dtype == TG_BF16) hipLaunchKernelGGL((pack_weights_kernel<u16>), grid, dim3(256), 0, ST(stream), src_base, (u16*)dst_base, (u16*)nullptr, tab, transpose ? 1 : 0);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

// both compute copies ([tap][out][in] and [tap][in][out]) of every weight of the flat parameter buffer in ONE launch
extern "C" int tg_pack_weights_both(const float* src_base, void* dst_t, void* dst_n, int dst_dtype, const int64_t* tab, int count,
                                    void* stream) {
  TG_CHECK_ARG(src_base && dst_t && dst_n && tab && count > 0, "bad argument");
  dim3 grid(64, count, 2);
  if (dst_dtype == TG_F32) hipLaunchKernelGGL((pack_weights_kernel<float>), grid, dim3(256), 0, ST(stream), src_base, (float*)dst_t, (float*)dst_n, tab, -1);
  else if (dst_dtype == TG_BF16) hipLaunchKernelGGL((pack_weights_kernel<u16>), grid, dim3(256), 0, ST(stream), src_base, (u16*)dst_t, (u16*)dst_n, tab, -1);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

// fragment-order bf16 copies (forward and input-gradient operand) of `count` 64 -> 64 3x3 weights; tab: 2 x int64 per tensor
// (src offset in the flat fp32 buffer, dst offset in elements, 36864 per tensor, input channels)
extern "C" int tg_pack_weights_frag(const float* src_base, void* dst_t, void* dst_n, const int64_t* tab, int count, void* stream) {
  TG_CHECK_ARG(src_base && dst_t && dst_n && tab && count > 0, "bad argument");
  hipLaunchKernelGGL(pack_weights_frag_kernel, dim3(8, count, 2), dim3(256), 0, ST(stream), src_base, (u16*)dst_t, (u16*)dst_n, tab);
  TG_CHECK_LAUNCH();
}

extern "C" int tg_sum_sq_diff(const void* a, const void* b, int dtype, int64_t n, float scale, float* out,
                              void* stream) {
  TG_CHECK_ARG(a && b && out && n > 0, "bad argument");
  dim3 grid(grid_1d(n, 256 * 8, 1024));
  if (dtype == TG_F32) hipLaunchKernelGGL((sum_diff_kernel<float, 0>), TG_DET_GRID(grid), dim3(256), 0, ST(stream), (const float*)a, (const float*)b, n, scale, out);
  else if (dtype == TG_BF16) hipLaunchKernelGGL((sum_diff_kernel<u16, 0>), TG_DET_GRID(grid), dim3(256), 0, ST(stream), (const u16*)a, (const u16*)b, n, scale, out);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

extern "C" int tg_sum_abs_diff(const void* a, const void* b, int dtype, int64_t n, float scale, float* out,
                               void* stream) {
  TG_CHECK_ARG(a && b && out && n > 0, "bad argument");
  dim3 grid(grid_1d(n, 256 * 8, 1024));
  if (dtype == TG_F32) hipLaunchKernelGGL((sum_diff_kernel<float, 1>), TG_DET_GRID(grid), dim3(256), 0, ST(stream), (const float*)a, (const float*)b, n, scale, out);
  else if (dtype == TG_BF16) hipLaunchKernelGGL((sum_diff_kernel<u16, 1>), TG_DET_GRID(grid), dim3(256), 0, ST(stream), (const u16*)a, (const u16*)b, n, scale, out);
  else TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
#include <stdarg.h>
static thread_local char g_err[512] = "";
void tg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* tg_last_error_string(void) { return g_err; }
extern "C" int tg_version(void) { return 1; }

// ------------------------------------------------------------------------------------------------
// Data step BEFORE the path: tf_data_gaussDownby4 (lib/ops.py:347-367) -- depthwise k x k Gaussian, stride 4,
// VALID -- fused with the training loader's 4-pixel... `border`-pixel crop + preprocess of the HR target
// (lib/dataloader.py:306-332: target = preprocess(HR[border:-border]), input = preprocessLR(gauss_down(HR))).
// One thread per LR pixel: its k x k window (81 taps x 3 channels, fp32, L1/L2 resident) and the 4x4 HR block it maps to.
// No MFMA: 3 channels, memory-bound (HR read once: 12 B / HR pixel).
struct GaussP {
  float w[121];          // up to 11 x 11 taps (sigma 1.5 -> 9 x 9), row-major
};

__global__ __launch_bounds__(256) void gauss_down4_kernel(const float* __restrict__ hr, float* __restrict__ lr,
                                                          float* __restrict__ target, int N, int H, int W, int k, int border,
                                                          GaussP g) {
  const int ho = (H - k) / 4 + 1, wo = (W - k) / 4 + 1;
  const int n = N * ho * wo;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int j = e % wo, i = (e / wo) % ho, b = e / (wo * ho);
    const float* __restrict__ src = hr + ((int64_t)(b * H + 4 * i) * W + 4 * j) * 3;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) {
        const float wv = g.w[ky * k + kx];
        const float* __restrict__ p = src + ((int64_t)ky * W + kx) * 3;
        acc[0] += wv * p[0];
        acc[1] += wv * p[1];
        acc[2] += wv * p[2];
      }
    lr[(int64_t)e * 3] = acc[0];
    lr[(int64_t)e * 3 + 1] = acc[1];
    lr[(int64_t)e * 3 + 2] = acc[2];
    if (target) {                                            // preprocess(x) = 2x - 1 of the cropped HR block (lib/ops.py:13-16)
      const int Ht = 4 * ho, Wt = 4 * wo;
#pragma unroll
      for (int dy = 0; dy < 4; ++dy) {
        const float* __restrict__ s = hr + ((int64_t)(b * H + border + 4 * i + dy) * W + border + 4 * j) * 3;
        float* __restrict__ d = target + ((int64_t)(b * Ht + 4 * i + dy) * Wt + 4 * j) * 3;
#pragma unroll
        for (int q = 0; q < 12; ++q) d[q] = s[q] * 2.f - 1.f;
      }
    }
  }
}

extern "C" int tg_gauss_down4_preprocess(const float* hr, float* lr, float* target, int N, int H, int W, int k,
                                         const float* weights, int border, void* stream) {
  TG_CHECK_ARG(hr && lr && weights && N > 0 && k >= 1 && k <= 11 && H >= k && W >= k, "bad argument");
  const int ho = (H - k) / 4 + 1, wo = (W - k) / 4 + 1;
  TG_CHECK_ARG(!target || (border >= 0 && border + 4 * ho <= H && border + 4 * wo <= W), "target crop leaves the image");
  TG_CHECK_ARG((int64_t)N * H * W * 3 < ((int64_t)1 << 31), "tensor too large");
  GaussP g;
  for (int t = 0; t < k * k; ++t) g.w[t] = weights[t];
  hipLaunchKernelGGL(gauss_down4_kernel, dim3(grid_1d((int64_t)N * ho * wo, 256, 1 << 16)), dim3(256), 0, ST(stream), hr, lr,
                     target, N, H, W, k, border, g);
  TG_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------
// Output step AFTER the path: save_img's clip(img * 255, 0, 255).astype(uint8) -- truncation -- and the
// RGB -> BGR flip cv.imwrite wants (lib/ops.py:521-523, main.py:262-267), on the device: the host copy shrinks 4x
// (1 byte instead of 4 per value) and can run asynchronously into pinned memory (tecogan_amd/output.py).
__global__ __launch_bounds__(256) void frame_to_u8_kernel(const float* __restrict__ x, unsigned char* __restrict__ out,
                                                          int64_t npix, int bgr) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
    const float r = x[p * 3], g = x[p * 3 + 1], b = x[p * 3 + 2];
    const unsigned char cr = (unsigned char)fminf(fmaxf(r * 255.f, 0.f), 255.f);
    const unsigned char cg = (unsigned char)fminf(fmaxf(g * 255.f, 0.f), 255.f);
    const unsigned char cb = (unsigned char)fminf(fmaxf(b * 255.f, 0.f), 255.f);
    out[p * 3] = bgr ? cb : cr;
    out[p * 3 + 1] = cg;
    out[p * 3 + 2] = bgr ? cr : cb;
  }
}

extern "C" int tg_frame_to_u8(const float* frame, unsigned char* out, int64_t npix, int bgr, void* stream) {
  TG_CHECK_ARG(frame && out && npix > 0, "bad argument");
  hipLaunchKernelGGL(frame_to_u8_kernel, dim3(grid_1d(npix, 256, 1 << 16)), dim3(256), 0, ST(stream), frame, out, npix, bgr);
  TG_CHECK_LAUNCH();
}
