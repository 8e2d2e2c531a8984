// Weight-gradient kernel for the convolution engine (gfx950, fp32 MFMA, split-K + fp32 atomics).
//
// For the gather-form convolution described by the descriptor (X gathered, Y per output pixel):
//   dW[tap][cx][cy] += sum_m X[m@tap][cx] * Y[m][cy]       GEMM: M'=Cx, N'=Cy, K'=output pixels
//   dbias[cy]       += sum_m Y[m][cy]
// Covers slim.conv2d (X = layer input, Y = dOut -> HWIO) and slim.conv2d_transpose
// (X = dOut, Y = layer input -> [kh,kw,Cout,Cin]) of reference lib/ops.py:35-56.
//
// Both operands arrive pixel-major (NHWC rows), which is exactly the K-major layout the
// 16x16x4 fp32 MFMA wants for A^T/B: lane (i=lane&15, g=lane>>4) reads Xs[4g+j][i] -- 64 lanes hit
// 64 distinct banks with a 68-float row pitch.  bf16 operands are widened to fp32 while staging
// (weight gradients are accumulated in full fp32 products; the flat fp32 gradient buffer is also
// the RCCL all-reduce buffer).  grid = (taps, Cx/64 * Cy/64 tiles, pixel chunks).
#include "common.h"
#include <mutex>
#include <math.h>
#include <stdlib.h>

int tg_wgrad_bf16_try(const tg_conv_desc* d, const void* x, int x_dtype, int ldx, const void* y, int y_dtype, int ldy,
                      float* dw, float* dbias, hipStream_t st);

struct WgradP {
  const void* x;
  const void* y;
  float* dw;
  float* dbias;
  int N, Hx, Wx, Cx, Hy, Wy, Cy, KH, KW, s, pt, pl;
  int M, chunk;     // total output pixels, pixels per block (multiple of 32)
  int ytiles;
  int ldx, ldy;     // channel strides of X / Y in memory (>= Cx / Cy: zero-padded buffers)
  int vecx, vecy;   // 4-element vector loads allowed (stride % 4 == 0 and 16/8-byte aligned base)
};

template <typename T> struct LoadVec4;
template <> struct LoadVec4<float> {
  static __device__ __forceinline__ float4 ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
};
template <> struct LoadVec4<u16> {
  static __device__ __forceinline__ float4 ld(const u16* p) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
  }
};

template <typename TX, typename TY>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradP p) {
  constexpr int PITCH = 68;  // floats per LDS row (64 + 4): (4*PITCH) % 32 == 16
  __shared__ __attribute__((aligned(16))) float Xs[32 * PITCH];
  __shared__ __attribute__((aligned(16))) float Ys[32 * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tap = blockIdx.x, kh = tap / p.KW, kw = tap % p.KW;
  const int xt = blockIdx.y / p.ytiles, yt = blockIdx.y % p.ytiles;
  const int cx0 = xt * 64, cy0 = yt * 64;
  const int mbeg = blockIdx.z * p.chunk;
  const int mend = min(mbeg + p.chunk, p.M);
  const bool do_bias = p.dbias != nullptr && tap == 0 && xt == 0;

  const TX* __restrict__ gx = static_cast<const TX*>(p.x);
  const TY* __restrict__ gy = static_cast<const TY*>(p.y);

  // staging assignment: item = (row in 0..31, 4-channel group in 0..15); two items per thread
  const int srow = tid >> 4, sgrp = tid & 15;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  const int frow = lane & 15, fg = lane >> 4;

  for (int mb = mbeg; mb < mend; mb += 32) {
    // ---- stage 32 pixels x 64 channels of X (gathered at this tap) and Y -------------
    // One vector load per item from a clamped address, zeroed by select (no branch around the load: hipcc
    // would otherwise wait vmcnt(0) per load); the scalar path only serves channel tails / odd strides.
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = srow + h * 16;
      const int m = min(mb + r, mend - 1);
      const bool mok = mb + r < mend;
      const int ox = m % p.Wy, t = m / p.Wy;
      const int oy = t % p.Hy, n = t / p.Hy;
      const int iy = oy * p.s - p.pt + kh, ix = ox * p.s - p.pl + kw;
      const int cx = cx0 + sgrp * 4, cy = cy0 + sgrp * 4;
      const bool xok = mok && iy >= 0 && iy < p.Hx && ix >= 0 && ix < p.Wx && cx < p.Cx;
      const bool yok = mok && cy < p.Cy;
      const int64_t xoff = xok ? ((int64_t)(n * p.Hx + iy) * p.Wx + ix) * p.ldx + cx : 0;
      const int64_t yoff = yok ? (int64_t)m * p.ldy + cy : 0;
      float4 vx, vy;
      if (p.vecx && cx + 3 < p.Cx) {
        vx = LoadVec4<TX>::ld(gx + xoff);
      } else {
        const int nx = xok ? min(p.Cx - cx, 4) : 0;
        vx.x = nx > 0 ? Elem<TX>::ld(gx + xoff) : 0.f;
        vx.y = nx > 1 ? Elem<TX>::ld(gx + xoff + 1) : 0.f;
        vx.z = nx > 2 ? Elem<TX>::ld(gx + xoff + 2) : 0.f;
        vx.w = nx > 3 ? Elem<TX>::ld(gx + xoff + 3) : 0.f;
      }
      if (p.vecy && cy + 3 < p.Cy) {
        vy = LoadVec4<TY>::ld(gy + yoff);
      } else {
        const int ny = yok ? min(p.Cy - cy, 4) : 0;
        vy.x = ny > 0 ? Elem<TY>::ld(gy + yoff) : 0.f;
        vy.y = ny > 1 ? Elem<TY>::ld(gy + yoff + 1) : 0.f;
        vy.z = ny > 2 ? Elem<TY>::ld(gy + yoff + 2) : 0.f;
        vy.w = ny > 3 ? Elem<TY>::ld(gy + yoff + 3) : 0.f;
      }
      if (!xok) vx = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!yok) vy = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(&Xs[r * PITCH + sgrp * 4]) = vx;
      *reinterpret_cast<float4*>(&Ys[r * PITCH + sgrp * 4]) = vy;
    }
    __syncthreads();
    if (do_bias && tid < 64) {
#pragma unroll 8
      for (int r = 0; r < 32; ++r) bsum += Ys[r * PITCH + tid];
    }
#pragma unroll
    for (int kk = 0; kk < 32; kk += 16) {
      float a[2][4], b[2][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = (kk + fg * 4 + e) * PITCH;
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i][e] = Xs[row + wm * 32 + i * 16 + frow];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j][e] = Ys[row + wn * 32 + j * 16 + frow];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  float* __restrict__ dw = p.dw + (int64_t)tap * p.Cx * p.Cy;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cx = cx0 + wm * 32 + i * 16 + fg * 4 + r;
        const int cy = cy0 + wn * 32 + j * 16 + frow;
        if (cx < p.Cx && cy < p.Cy) unsafeAtomicAdd(dw + (int64_t)cx * p.Cy + cy, acc[i][j][r]);
      }
  if (do_bias && tid < 64 && cy0 + tid < p.Cy) unsafeAtomicAdd(p.dbias + cy0 + tid, bsum);
}

extern "C" int tg_conv_wgrad(const tg_conv_desc* d, const void* x, int x_dtype, int ldx, const void* y, int y_dtype,
                             int ldy, float* dw, float* dbias, void* stream) {
  TG_CHECK_ARG(d && x && y && dw, "null pointer");
  TG_CHECK_ARG(d->mode == 0, "descriptor must be the gather form");
  TG_CHECK_ARG(d->stride >= 1 && d->stride <= 2, "stride must be 1 or 2");
  TG_CHECK_ARG((int64_t)d->N * d->Hout * d->Wout < (1ll << 31), "too many pixels");
  WgradP p;
  p.x = x; p.y = y; p.dw = dw; p.dbias = dbias;
  p.N = d->N; p.Hx = d->Hin; p.Wx = d->Win; p.Cx = d->Cin;
  p.Hy = d->Hout; p.Wy = d->Wout; p.Cy = d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.s = d->stride; p.pt = d->pad_t; p.pl = d->pad_l;
  p.M = d->N * d->Hout * d->Wout;
  p.ldx = ldx > 0 ? ldx : d->Cin;
  p.ldy = ldy > 0 ? ldy : d->Cout;
  TG_CHECK_ARG(p.ldx >= d->Cin && p.ldy >= d->Cout, "channel stride smaller than channel count");
  p.vecx = p.ldx % 4 == 0 && ((uintptr_t)x % 16 == 0);
  p.vecy = p.ldy % 4 == 0 && ((uintptr_t)y % 16 == 0);
  if (tg_wgrad_bf16_try(d, x, x_dtype, p.ldx, y, y_dtype, p.ldy, dw, dbias, static_cast<hipStream_t>(stream)))
    TG_CHECK_LAUNCH();
  const int xtiles = (p.Cx + 63) / 64;
  p.ytiles = (p.Cy + 63) / 64;
  const int base_blocks = d->KH * d->KW * xtiles * p.ytiles;
  // aim for ~2048 blocks, at least 64 pixels of reduction per block
  int ksplit = (2048 + base_blocks - 1) / base_blocks;
  const int max_split = (p.M + 63) / 64;
  if (ksplit > max_split) ksplit = max_split;
  if (ksplit < 1 || tg_det()) ksplit = 1;                  // parity mode: no split-K (one add per dW element and launch)
  p.chunk = (((p.M + ksplit - 1) / ksplit) + 31) / 32 * 32;
  ksplit = (p.M + p.chunk - 1) / p.chunk;
  dim3 grid(d->KH * d->KW, xtiles * p.ytiles, ksplit);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double wfl = 2.0 * (double)p.M * d->KH * d->KW * (double)d->Cin * d->Cout;
  const double wby = (double)d->N * d->Hin * d->Win * p.ldx * (x_dtype == TG_F32 ? 4 : 2) +
                     (double)p.M * p.ldy * (y_dtype == TG_F32 ? 4 : 2) + 4.0 * d->KH * d->KW * d->Cin * d->Cout;
  if (x_dtype == TG_F32 && y_dtype == TG_F32)
    TG_LAUNCH("conv_wgrad<f32,f32>", wfl, wby, (conv_wgrad_kernel<float, float>), grid, dim3(256), 0, st, p);
  else if (x_dtype == TG_BF16 && y_dtype == TG_BF16)
    TG_LAUNCH("conv_wgrad<bf16,bf16>", wfl, wby, (conv_wgrad_kernel<u16, u16>), grid, dim3(256), 0, st, p);
  else if (x_dtype == TG_BF16 && y_dtype == TG_F32)
    TG_LAUNCH("conv_wgrad<bf16,f32>", wfl, wby, (conv_wgrad_kernel<u16, float>), grid, dim3(256), 0, st, p);
  else if (x_dtype == TG_F32 && y_dtype == TG_BF16)
    TG_LAUNCH("conv_wgrad<f32,bf16>", wfl, wby, (conv_wgrad_kernel<float, u16>), grid, dim3(256), 0, st, p);
  else
    TG_CHECK_ARG(false, "bad dtype");
  TG_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------
// column sum: out[c] += sum_rows x[row][c]   (bias gradient of conv_transpose, BN beta, ...)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t rows, int C,
                                                     float* __restrict__ out) {
  // block handles a strip of rows; thread t handles channel (t % Cb) for rows stepping by 256/Cb
  __shared__ float red[256];
  const int Cb = C < 256 ? C : 256;
  const int lanes_per_c = 256 / Cb;          // rows processed in parallel per channel
  for (int c0 = blockIdx.y * Cb; c0 < C; c0 += gridDim.y * Cb) {
    const int c = c0 + (threadIdx.x % Cb);
    const int rsub = threadIdx.x / Cb;
    float s = 0.f;
    if (rsub < lanes_per_c && c < C)
      for (int64_t r = (int64_t)blockIdx.x * lanes_per_c + rsub; r < rows; r += (int64_t)gridDim.x * lanes_per_c)
        s += Elem<T>::ld(x + r * C + c);
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < Cb && c < C) {
      float t = 0.f;
      for (int k = 0; k < lanes_per_c; ++k) t += red[k * Cb + threadIdx.x];
      unsafeAtomicAdd(out + c, t);
    }
    __syncthreads();
  }
}

// bf16, C a multiple of 8 with C/8 a power of two <= 256: every thread streams 16-byte channel octets (8 fp32
// partial sums), four row-strides in flight; one LDS reduction and C atomics per workgroup.
__global__ __launch_bounds__(256) void colsum_bf16x8_kernel(const u16* __restrict__ x, int64_t rows, int C,
                                                            float* __restrict__ out) {
  __shared__ float red[256 * 8];
  const int OC = C >> 3, RP = 256 / OC;
  const int oc = threadIdx.x % OC, rsub = threadIdx.x / OC;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  const int64_t stride = (int64_t)gridDim.x * RP;
  int64_t r = (int64_t)blockIdx.x * RP + rsub;
  auto accum = [&](const uint4& v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s[2 * e] += __uint_as_float(w[e] << 16);
      s[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
    }
  };
  for (; r + 3 * stride < rows; r += 4 * stride) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const uint4*>(x + (r + k * stride) * C + oc * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) accum(v[k]);
  }
  for (; r < rows; r += stride) accum(*reinterpret_cast<const uint4*>(x + r * C + oc * 8));
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[e];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int k = 0; k < RP; ++k) t += red[(k * OC + (c >> 3)) * 8 + (c & 7)];
    unsafeAtomicAdd(out + c, t);
  }
}

extern "C" int tg_colsum(const void* x, int dtype, int64_t rows, int C, float* out, void* stream) {
  TG_CHECK_ARG(x && out && rows > 0 && C > 0, "bad argument");
  if (dtype == TG_BF16 && C % 8 == 0 && C / 8 <= 256 && ((C / 8) & (C / 8 - 1)) == 0 && ((uintptr_t)x & 15) == 0) {
    const int RP = 256 / (C / 8);
    int gx = (int)cdiv64(rows, (int64_t)RP * 8);
    if (gx > 256) gx = 256;              // every workgroup ends with C atomics on the same C addresses
    hipLaunchKernelGGL(colsum_bf16x8_kernel, TG_DET_GRID(gx), dim3(256), 0, static_cast<hipStream_t>(stream), (const u16*)x, rows,
                       C, out);
    TG_CHECK_LAUNCH();
  }
  const int Cb = C < 256 ? C : 256;
  const int lpc = 256 / Cb;
  int gx = (int)cdiv64(rows, (int64_t)lpc * 64);
  if (gx > 512) gx = 512;
  if (gx < 1) gx = 1;
  dim3 grid(tg_det() ? 1 : gx, (C + Cb - 1) / Cb);          // (parity mode: one workgroup per channel block)
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == TG_F32) hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, st, (const float*)x, rows, C, out);
  else hipLaunchKernelGGL((colsum_kernel<u16>), grid, dim3(256), 0, st, (const u16*)x, rows, C, out);
  TG_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------
// bf16 MFMA weight gradient (throughput mode: both operands bf16, 16-byte aligned channel strides).
// dW[tap][cx][cy] += sum_pix X[pix@tap][cx] * Y[pix][cy] with v_mfma_f32_16x16x32_bf16: the reduction index
// (pixels) must be the contiguous one inside a fragment, the opposite of NHWC.  Each thread therefore loads one
// pixel PAIR x 8 channels (two 16-byte loads) and writes 8 packed {pixel p, pixel p+1} dwords into a transposed
// LDS panel Xt[channel][64 pixels]; fragments are then two ds_read_b64 per 8 pixels.  Panel pitch 136 B keeps the
// transposing writes conflict-free (8 rows * 34 dwords = 16 mod 32).  Waves 0-1 stage X, waves 2-3 stage Y.
// K-step = 64 pixels; each wave owns a 32x32 quadrant of the 64x64 (cx, cy) tile; fp32 accumulate; split-K over
// pixel chunks with fp32 atomics into the flat gradient buffer.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Split-K degree.  Measured model of these kernels on MI355X (tools/mb_wgrad.py sweeps + tools/trace_wgrad.py):
//   T ~ fixed + a * (64-pixel steps per workgroup) + b * (fp32 atomics issued, in millions)
// with a ~ 0.85 us (a step is issue-latency bound at the 1-2 waves per SIMD these launches run at) and b ~ 2 us
// (the L2 atomic units retire ~0.5 T adds/s chip-wide).  steps = M/64/ksplit, atomics = outputs * ksplit:
// the optimum is ksplit = sqrt(a * (M/64) / (b * outputs)), capped so that the grid stays within ~1024 workgroups.
static int tg_wgrad_ksplit(int M, int64_t outputs, int base_blocks, int quantum) {
  const int target_env = 0;
  int ksplit;
  if (target_env) {
    ksplit = (target_env + base_blocks - 1) / base_blocks;
  } else {
    const double steps = (double)M / 64.0, b = 2.0e-6 * (double)outputs;
    ksplit = (int)(sqrt(0.85 * steps / b) + 0.5);
    const int cap = (1024 + base_blocks - 1) / base_blocks;
    if (ksplit > cap) ksplit = cap;
  }
  const int max_split = (M + 2 * quantum - 1) / (2 * quantum);
  if (ksplit > max_split) ksplit = max_split;
  if (ksplit < 1 || tg_det()) ksplit = 1;                  // parity mode: no split-K
  return ksplit;
}

struct WgradBP {
  const u16* x;
  const u16* y;
  float* dw;
  float* dbias;
  int N, Hx, Wx, Cx, Hy, Wy, Cy, KH, KW, s, pt, pl;
  int M, chunk, xtiles, ytiles, ldx, ldy;
  unsigned xbytes, ybytes;             // buffer extents for the bounds-checked loads
};

// Cycle-stamp instrumentation (tools/trace_wgrad.py builds a private copy of the library with -DTG_WGRAD_TRACE; the
// product build contains none of it): wave 0 of one mid-grid workgroup records s_memtime at the phase boundaries.
#ifdef TG_WGRAD_TRACE
__device__ unsigned long long tg_wgrad_trace_buf[64];
#define TG_STAMP(i)                                                                  \
  do {                                                                               \
    if (trace_on && (i) < 64) tg_wgrad_trace_buf[(i)] = (unsigned long long)clock64(); \
  } while (0)
extern "C" int tg_debug_wgrad_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tg_wgrad_trace_buf), sizeof(unsigned long long) * 64);
}
#else
#define TG_STAMP(i) do { } while (0)
#endif

// PF = 64-pixel slots per macro-step: one barrier pair covers 64*PF pixels.  A cycle-stamp trace (tools/trace_wgrad.py)
// of the 64-pixel version showed ~1700 cycles per step around 128 cycles of MFMA at the 1-2 waves per SIMD these
// launches run at: the stage -> barrier -> fragment reads -> MFMA -> barrier chain is pure exposed latency, so the fix is
// more pixels per trip through it (and 4*PF loads in flight across the MFMA block).  The host makes `chunk` a multiple of
// 64*PF; slots past the end run on zeros (out-of-range lanes fetch nothing).
// XCD-aware work mapping.  Workgroup b runs on XCD b % 8, each XCD has its own L2, and the KH*KW taps (and channel
// tiles) of one pixel chunk all read the same X / Y rows.  With the natural (tap, tile, chunk) grid the taps of a
// chunk were sprayed over all 8 XCDs, so every L2 pulled the whole of X and Y through the fabric (8x the traffic:
// ~84 MB per launch at the generator shape, which is what bounded the kernel at ~20 us).  Bijective remap: XCD x owns a
// contiguous range of work items, a work item = (chunk, tile, tap) with tap fastest.
__device__ __forceinline__ int tg_xcd_work_item() {
  const int nwg = gridDim.x, L = blockIdx.x;
  const int xcd = L & 7, slot = L >> 3, q8 = nwg >> 3, r8 = nwg & 7;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
}

// body of one work item `work` = (pixel chunk, channel tile, tap) of the layer described by p
template <int PF>
__device__ __forceinline__ void wgrad_bf16_body(const WgradBP& p, const int work, [[maybe_unused]] const int nwg) {
  constexpr int ROWB = 128 * PF + 8;             // bytes per channel row: 64*PF pixels * 2 B + 8 pad
  extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
  unsigned char* Xt = wg_smem;                    // [64 channels][ROWB]
  unsigned char* Yt = wg_smem + 64 * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
#ifdef TG_WGRAD_TRACE
  const bool trace_on = work == (nwg / 2) && threadIdx.x == 0;
  int stamp = 3;
#endif
  TG_STAMP(0);
  const int per_chunk = p.KH * p.KW * p.xtiles * p.ytiles;
  const int zc = work / per_chunk, rem = work - zc * per_chunk;
  const int ntap = p.KH * p.KW, tile = rem / ntap, tap = rem - tile * ntap;
  const int kh = tap / p.KW, kw = tap % p.KW;
  const int xt = tile / p.ytiles, yt = tile % p.ytiles;
  const int cx0 = xt * 64, cy0 = yt * 64;
  const int mbeg = zc * p.chunk, mend = min(mbeg + p.chunk, p.M);
  const bool do_bias = p.dbias != nullptr && tap == 0 && xt == 0;
  const int frow = lane & 15, fg = lane >> 4;
  // staging role: threads 0..127 -> X, 128..255 -> Y; item = (pixel pair 0..31, channel octet 0..7)... 256 items each,
  // two items per thread (pairs pp and pp+16)
  const bool stage_x = wave < 2;                   // wave-uniform: every role parameter below is scalar
  const int st = tid & 127, oct = st >> 4, pp0 = st & 15;
  // One address formula for both roles: Y is the "1x1, stride 1, no padding" case of the X gather.  (A per-lane
  // `stage_x ? p.x : p.y` made hipcc fetch the pointer from the kernarg segment with a vector load and wait
  // vmcnt(0) in front of EVERY data load -- the whole burst was serialised.)
  // Loads are buffer loads through a wave-uniform descriptor: a lane that must read zero (padding, tail) gets an
  // out-of-range offset and the hardware returns 0.  A select AFTER the load (`if (!ok) v = 0`) sits in the
  // issuing basic block and made every prefetch wait for its own data before the MFMAs started.
  const unsigned long long sbase = (unsigned long long)(stage_x ? p.x : p.y);
  const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)sbase);
  const unsigned bhi = __builtin_amdgcn_readfirstlane((unsigned)(sbase >> 32));
  const unsigned sbytes = __builtin_amdgcn_readfirstlane(stage_x ? p.xbytes : p.ybytes);
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)bhi << 32) | blo), 0, (int)sbytes,
                                                      0x00020000);
  const int Hs = stage_x ? p.Hx : p.Hy, Ws = stage_x ? p.Wx : p.Wy, lds = stage_x ? p.ldx : p.ldy;
  const int strd = stage_x ? p.s : 1, offy = stage_x ? kh - p.pt : 0, offx = stage_x ? kw - p.pl : 0;
  const int cbase = (stage_x ? cx0 : cy0) + oct * 8;
  const bool cok = cbase < lds;
  // Incremental addressing: the K loop advances every thread's pixel by 64; (ix, iy, byte offset) follow with adds and
  // two conditional corrections (column carry, image wrap).  The first version recomputed n/oy/ox per load with
  // magic-number divisions: 44 quarter-rate integer multiplies per step made the loop VALU-bound (~1000 cycles of
  // address math against 128 cycles of MFMA).
  const int WyS = p.Wy * strd, HyS = p.Hy * strd;
  const int ixLim = WyS + offx, iyLim = HyS + offy;
  const int q64 = 64 / p.Wy, r64 = 64 - q64 * p.Wy, qH = q64 / p.Hy, rH = q64 - qH * p.Hy;
  const int dxs = r64 * strd, dys = rH * strd;
  const unsigned K0 = (unsigned)((dxs + dys * Ws + qH * Hs * Ws) * lds * 2);
  const unsigned K1 = (unsigned)((strd * Ws - WyS) * lds * 2);           // column carry: ox -= Wy, oy += 1
  const unsigned K2 = (unsigned)((Hs * Ws - HyS * Ws) * lds * 2);        // image wrap:   oy -= Hy, n += 1
  const unsigned qstep = (unsigned)(strd * lds * 2);                      // pixel m+1 (same row: m and Wy are even)
  int sm[2], six[2], siy[2];
  unsigned soff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int m = mbeg + (pp0 + h * 16) * 2;
    const int t = m / p.Wy, ox = m - t * p.Wy;
    const int n = t / p.Hy, oy = t - n * p.Hy;
    sm[h] = m;
    six[h] = ox * strd + offx;
    siy[h] = oy * strd + offy;
    soff[h] = (unsigned)(((n * Hs + siy[h]) * Ws + six[h]) * lds + cbase) * 2u;
  }

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient (column sums of Y): the Y-staging threads of the tap-0 workgroups add up the 8 channels x 4 pixels
  // they hold in registers anyway.  (Summing the staged LDS rows instead -- 32 dependent reads per step in one wave --
  // put ~1000 cycles per step on the critical path of those workgroups: 5 of 22 us at the generator shape.)
  const bool bias_thread = do_bias && !stage_x;
  float bsum[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bsum[k] = 0.f;

  uint4 v[PF][2][2];
  auto load_block = [&](uint4 (&dst)[2][2]) {   // issues the 4 loads of the current state, then advances it by 64 pixels
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool rowok = cok & ((unsigned)siy[h] < (unsigned)Hs);
      const bool ok0 = rowok & (sm[h] < mend) & ((unsigned)six[h] < (unsigned)Ws);
      const bool ok1 = rowok & (sm[h] + 1 < mend) & ((unsigned)(six[h] + strd) < (unsigned)Ws);
      const u32x4 t0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(ok0 ? soff[h] : 0x80000000u), 0, 0);
      const u32x4 t1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(ok1 ? soff[h] + qstep : 0x80000000u), 0, 0);
      dst[h][0] = make_uint4(t0.x, t0.y, t0.z, t0.w);
      dst[h][1] = make_uint4(t1.x, t1.y, t1.z, t1.w);
      sm[h] += 64;
      six[h] += dxs;
      siy[h] += dys;
      soff[h] += K0;
      const bool c = six[h] >= ixLim;
      six[h] -= c ? WyS : 0;
      siy[h] += c ? strd : 0;
      soff[h] += c ? K1 : 0u;
      const bool w = siy[h] >= iyLim;
      siy[h] -= w ? HyS : 0;
      soff[h] += w ? K2 : 0u;
    }
  };
  TG_STAMP(1);
#pragma unroll
  for (int d = 0; d < PF; ++d) load_block(v[d]);
  TG_STAMP(2);
  unsigned char* panel = stage_x ? Xt : Yt;
  for (int mb = mbeg; mb < mend; mb += 64 * PF) {
#ifdef TG_WGRAD_TRACE
    TG_STAMP(stamp); ++stamp;          // macro-step top
#endif
    if (bias_thread) {                 // wave-uniform branch
#pragma unroll
      for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint32_t w4[4] = {v[d][h][q].x, v[d][h][q].y, v[d][h][q].z, v[d][h][q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              bsum[2 * e] += __uint_as_float(w4[e] << 16);
              bsum[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u);
            }
          }
    }
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t* a = reinterpret_cast<const uint32_t*>(&v[d][h][0]);   // pixel 2pp   : channels 8*oct .. +7
        const uint32_t* b = reinterpret_cast<const uint32_t*>(&v[d][h][1]);   // pixel 2pp+1
        const int pp = pp0 + h * 16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // channels 2e, 2e+1 of the octet: {lo = pixel 2pp, hi = pixel 2pp+1}
          const uint32_t c0 = (a[e] & 0xffffu) | (b[e] << 16);
          const uint32_t c1 = (a[e] >> 16) | (b[e] & 0xffff0000u);
          *reinterpret_cast<uint32_t*>(panel + (oct * 8 + 2 * e) * ROWB + d * 128 + pp * 4) = c0;
          *reinterpret_cast<uint32_t*>(panel + (oct * 8 + 2 * e + 1) * ROWB + d * 128 + pp * 4) = c1;
        }
      }
#ifdef TG_WGRAD_TRACE
    TG_STAMP(stamp); ++stamp;          // data arrived + staged
#endif
    __syncthreads();
#ifdef TG_WGRAD_TRACE
    TG_STAMP(stamp); ++stamp;          // barrier passed
#endif
#pragma unroll
    for (int d = 0; d < PF; ++d) load_block(v[d]);   // next macro-step, unconditionally (lanes past the end fetch
                                                     // nothing): 4*PF loads in flight across the MFMA block below
#pragma unroll
    for (int kk = 0; kk < 2 * PF; ++kk) {            // 2*PF x 32 pixels
      bf16x8 af[2], bfm[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned char* r = Xt + (wm * 32 + i * 16 + frow) * ROWB + kk * 64 + fg * 16;
        uint2 lo = *reinterpret_cast<const uint2*>(r), hi = *reinterpret_cast<const uint2*>(r + 8);
        uint4 w4 = make_uint4(lo.x, lo.y, hi.x, hi.y);
        af[i] = *reinterpret_cast<bf16x8*>(&w4);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned char* r = Yt + (wn * 32 + j * 16 + frow) * ROWB + kk * 64 + fg * 16;
        uint2 lo = *reinterpret_cast<const uint2*>(r), hi = *reinterpret_cast<const uint2*>(r + 8);
        uint4 w4 = make_uint4(lo.x, lo.y, hi.x, hi.y);
        bfm[j] = *reinterpret_cast<bf16x8*>(&w4);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfm[j], acc[i][j], 0, 0, 0);
    }
#ifdef TG_WGRAD_TRACE
    TG_STAMP(stamp); ++stamp;          // MFMAs issued
#endif
    __syncthreads();
  }
#ifdef TG_WGRAD_TRACE
  TG_STAMP(60);
#endif
  float* __restrict__ dw = p.dw + (int64_t)tap * p.Cx * p.Cy;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cx = cx0 + wm * 32 + i * 16 + fg * 4 + r;
        const int cy = cy0 + wn * 32 + j * 16 + frow;
        if (cx < p.Cx && cy < p.Cy) unsafeAtomicAdd(dw + (int64_t)cx * p.Cy + cy, acc[i][j][r]);
      }
  if (do_bias) {                       // workgroup-uniform
    // 16 lanes (pixel pairs) share a channel octet: xor-reduce, gather the 64 sums in LDS and let ONE instruction add
    // them (64 consecutive addresses).  Per-octet atomics from 16 separate instructions per workgroup serialised on the
    // four cache lines every bias workgroup hits: +7 us at the generator shape.
    float* red = reinterpret_cast<float*>(Xt);       // the panels are idle after the loop's final barrier
    if (bias_thread) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) bsum[k] += __shfl_xor(bsum[k], o, 64);
        if (pp0 == 0) red[oct * 8 + k] = bsum[k];
      }
    }
    __syncthreads();
    if (tid < 64 && cy0 + tid < p.Cy) unsafeAtomicAdd(p.dbias + cy0 + tid, red[tid]);
  }
#ifdef TG_WGRAD_TRACE
  __builtin_amdgcn_s_waitcnt(0);       // atomics retired (vmcnt/lgkmcnt/expcnt = 0)
  TG_STAMP(61);
#endif
}

template <int PF>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(WgradBP p) {
  wgrad_bf16_body<PF>(p, tg_xcd_work_item(), (int)gridDim.x);
}

// Several layers of DIFFERENT geometry in one launch (round 5: the discriminator's four stride-2 4x4 convs, reference
// lib/Teco.py:35-39,52-66 under tf.gradients): a launch of this kernel costs ~30 us before its first and after its last useful
// step whatever the layer's size (prologue, split-K atomics tail) -- the three small layers of D (1.6 - 3.2 GFLOP) cost as much
// as the large one (12.9) -- so the layers go out together and the small ones fill the large one's tail.  A workgroup finds its
// layer by scanning the work-prefix table (workgroup-uniform), then runs the single-layer body on that layer's descriptor.
#define TG_WGRAD_BM_MAX 8
struct WgradBM {
  WgradBP g[TG_WGRAD_BM_MAX];
  int wstart[TG_WGRAD_BM_MAX + 1];
  int groups;
};
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_multi_kernel(WgradBM P) {
  const int work = tg_xcd_work_item();
  int grp = 0;
  for (int g1 = 1; g1 < P.groups; ++g1) grp = work >= P.wstart[g1] ? g1 : grp;
  if (work >= P.wstart[P.groups]) return;
  const WgradBP p = P.g[grp];
  wgrad_bf16_body<2>(p, work - P.wstart[grp], (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// Row kernel: stride-1 convolutions with 3 horizontal taps and SAME width (every 3x3 layer of generator_F, FNet and the
// D input conv).  One workgroup accumulates the THREE kw taps of one kernel row kh for a 64x64 channel tile:
//   * the Y^T panel is staged once and its fragments feed 3x the MFMAs;
//   * the three shifted X pixel pairs {(x-1,x), (x,x+1), (x+1,x+2)} come from FOUR 16-byte loads of one input row;
//   * with stride 1 and equal extents the input offset is linear in the output pixel index, so the per-step address
//     update is one add (plus the (ox, iy) bookkeeping for the zero padding).
// The cycle-stamp trace of the per-tap kernel (tools/trace_wgrad.py) showed ~1300 issue cycles of staging / address /
// fragment-read instructions around 128 cycles of MFMA per 64-pixel step at the 1-2 waves per SIMD these launches get:
// the loop is instruction-issue bound, so the lever is MFMAs per staged byte.  Here: 24 MFMAs per step for about the same
// instruction count.  Panels are [64 channels][64 pixels] with a 144-byte pitch: conflict-free transposing writes for
// the (32 pixel pairs x 2 octets) a wave writes per instruction, and 16-byte aligned rows -> one ds_read_b128 per fragment.
// Grouped form: `groups` layers of identical geometry (the 2*num_resblock 64->64 convs of generator_F) in ONE launch,
// each with its own X / Y / dW / dbias pointers.  A launch of this kernel costs ~8 us before the first useful step
// (launch, prologue, first loads) and after the last (atomics drain); one launch for all layers pays that once, and
// with groups x more workgroups per launch the split-K degree per layer (= atomics) drops.
#define TG_WGRAD_MAX_GROUPS 40
#define TG_WGRAD_MAX_GEOS 16
// geometry of a group (layer): the grouped launch may mix layers of DIFFERENT geometry (round 3: FNet's 14 weight gradients
// in one launch); groups sharing a geometry (the generator trunk) share a table entry
struct WgradGeo {
  int N, H, W, Cx, Cy, KH, pt;
  int M, chunk, nchunk, xtiles, ytiles, ldx, ldy;
  unsigned xbytes, ybytes;
};
struct WgradRP {
  const u16* xs[TG_WGRAD_MAX_GROUPS];
  const u16* ys[TG_WGRAD_MAX_GROUPS];
  float* dws[TG_WGRAD_MAX_GROUPS];
  float* dbs[TG_WGRAD_MAX_GROUPS];
  int wstart[TG_WGRAD_MAX_GROUPS + 1];     // first work item of every group (prefix sums of KH * xtiles * ytiles * nchunk)
  unsigned char gidx[TG_WGRAD_MAX_GROUPS];  // geometry table entry of every group
  WgradGeo geo[TG_WGRAD_MAX_GEOS];
  int groups;
};

__device__ __forceinline__ void tg_interleave_store(const u32x4& a, const u32x4& b, unsigned char* dst) {
  // a = 8 channels of pixel 2pp, b = of pixel 2pp+1; row c gets {lo = a[c], hi = b[c]} (one v_perm_b32 per dword)
  const unsigned av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    *reinterpret_cast<unsigned*>(dst + (2 * e) * 144) = __builtin_amdgcn_perm(bv[e], av[e], 0x05040100u);
    *reinterpret_cast<unsigned*>(dst + (2 * e + 1) * 144) = __builtin_amdgcn_perm(bv[e], av[e], 0x07060302u);
  }
}

__global__ __launch_bounds__(256, 2) void conv_wgrad_row3_bf16_kernel(WgradRP p) {
  constexpr int ROWB = 144, PANEL = 64 * ROWB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * PANEL];   // X(kw=0), X(kw=1), X(kw=2), Y
  unsigned char* Yt = smem + 3 * PANEL;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fg = lane >> 4;
  // XCD-aware work mapping (see conv_wgrad_bf16_kernel): work item = (group, chunk, channel tile, kh), kh fastest
  const int nwg = gridDim.x, L = blockIdx.x;
  const int xcd = L & 7, slot = L >> 3, q8 = nwg >> 3, r8 = nwg & 7;
  const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  int grp = 0;                                        // workgroup-uniform scan of the (<= 40 entry) prefix table
  for (int g1 = 1; g1 < p.groups; ++g1) grp = work >= p.wstart[g1] ? g1 : grp;
  if (work >= p.wstart[p.groups]) return;
  const WgradGeo g = p.geo[p.gidx[grp]];
  const int wrem = work - p.wstart[grp];
  const int per_chunk = g.KH * g.xtiles * g.ytiles;
  const int zc = wrem / per_chunk, rem = wrem - zc * per_chunk;
  const u16* __restrict__ gx = p.xs[grp];             // workgroup-uniform: scalar loads from the kernarg tables
  const u16* __restrict__ gy = p.ys[grp];
  float* __restrict__ gdw = p.dws[grp];
  float* __restrict__ gdb = p.dbs[grp];
  const int tile = rem / g.KH, kh = rem - tile * g.KH;
  const int xt = tile / g.ytiles, yt = tile - xt * g.ytiles;
  const int cx0 = xt * 64, cy0 = yt * 64;
  const int mbeg = zc * g.chunk, mend = min(mbeg + g.chunk, g.M);
  const bool do_bias = gdb != nullptr && kh == 0 && xt == 0;

  const auto rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(gx), 0, (int)g.xbytes, 0x00020000);
  const auto rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(gy), 0, (int)g.ybytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;

  // staging item of this thread: pixel pair pp (pixels 2pp, 2pp+1 of the 64-pixel step) x channel octet oct, for X and Y
  const int pp = tid & 31, oct = tid >> 5;
  const bool cxok = cx0 + oct * 8 < g.ldx, cyok = cy0 + oct * 8 < g.ldy;
  int m = mbeg + 2 * pp;                                    // output pixel of the pair (even; W is even)
  int ox, iy;                                               // output column, input row of this kernel row
  {
    const int t = m / g.W;
    ox = m - t * g.W;
    iy = t % g.H + kh - g.pt;
  }
  // stride 1, equal extents: the input pixel of (output pixel m, tap kh, kw) is m + (kh-pt)*W + (kw-1), linear in m
  unsigned offx = (unsigned)(((m + (kh - g.pt) * g.W) * g.ldx + cx0 + oct * 8) * 2);
  unsigned offy = (unsigned)((m * g.ldy + cy0 + oct * 8) * 2);
  const unsigned pixb = (unsigned)(g.ldx * 2), stepx = 64u * pixb, stepy = (unsigned)(64 * g.ldy * 2);
  const int q64 = 64 / g.W, r64 = 64 - q64 * g.W, rH = q64 % g.H;
  const int iyLim = g.H + kh - g.pt;

  u32x4 rx[4], ry[2];
  auto load_step = [&]() {             // 6 loads for the current state, then advance the state by 64 pixels
    const bool rowok = cxok & ((unsigned)iy < (unsigned)g.H);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = rowok & ((unsigned)(ox - 1 + j) < (unsigned)g.W);
      rx[j] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)(ok ? offx + (unsigned)(j - 1) * pixb : OOB), 0, 0);
    }
    // pixels past the chunk contribute nothing because their Y is zero (X stays finite: in-image data or zeros)
    ry[0] = __builtin_amdgcn_raw_buffer_load_b128(rsy, (int)((cyok & (m < mend)) ? offy : OOB), 0, 0);
    ry[1] = __builtin_amdgcn_raw_buffer_load_b128(rsy, (int)((cyok & (m + 1 < mend)) ? offy + (unsigned)(g.ldy * 2) : OOB), 0, 0);
    m += 64;
    offx += stepx;
    offy += stepy;
    ox += r64;
    const bool c = ox >= g.W;
    ox -= c ? g.W : 0;
    iy += rH + (c ? 1 : 0);
    iy -= iy >= iyLim ? g.H : 0;
  };

  f32x4 acc[3][2][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bsum[k] = 0.f;

  unsigned char* wbase = smem + (oct * 8) * ROWB + pp * 4;
  const unsigned char* Abase = smem + (wm * 32 + frow) * ROWB + fg * 16;
  const unsigned char* Bbase = Yt + (wn * 32 + frow) * ROWB + fg * 16;

  load_step();
  for (int mb = mbeg; mb < mend; mb += 64) {
    if (do_bias) {                     // workgroup-uniform
      const unsigned w0[4] = {ry[0].x, ry[0].y, ry[0].z, ry[0].w}, w1[4] = {ry[1].x, ry[1].y, ry[1].z, ry[1].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bsum[2 * e] += __uint_as_float(w0[e] << 16) + __uint_as_float(w1[e] << 16);
        bsum[2 * e + 1] += __uint_as_float(w0[e] & 0xffff0000u) + __uint_as_float(w1[e] & 0xffff0000u);
      }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) tg_interleave_store(rx[t], rx[t + 1], wbase + t * PANEL);
    tg_interleave_store(ry[0], ry[1], wbase + 3 * PANEL);
    __syncthreads();
    load_step();                       // next step (lanes past the end fetch nothing): in flight across the MFMAs
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 bf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const uint4*>(Bbase + j * 16 * ROWB + kk * 64);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        uint4 af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const uint4*>(Abase + t * PANEL + i * 16 * ROWB + kk * 64);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&af[i]),
                                                                   *reinterpret_cast<bf16x8*>(&bf[j]), acc[t][i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float* __restrict__ dw = gdw + (int64_t)(kh * 3 + t) * g.Cx * g.Cy;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cx = cx0 + wm * 32 + i * 16 + fg * 4 + r;
          const int cy = cy0 + wn * 32 + j * 16 + frow;
          if (cx < g.Cx && cy < g.Cy) unsafeAtomicAdd(dw + (int64_t)cx * g.Cy + cy, acc[t][i][j][r]);
        }
  }
  if (do_bias) {                       // 32 lanes (pixel pairs) share an octet: xor-reduce, one 64-wide atomic instruction
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) bsum[k] += __shfl_xor(bsum[k], o, 64);
      if (pp == 0) red[oct * 8 + k] = bsum[k];
    }
    __syncthreads();
    if (tid < 64 && cy0 + tid < g.Cy) unsafeAtomicAdd(gdb + cy0 + tid, red[tid]);
  }
}

int tg_wgrad_tr_launch(const tg_conv_desc* d, int groups, const void* const* x, int ldx, const void* const* y, int ldy,
                       float* const* dw, float* const* dbias, hipStream_t st, const void* x_narrow = nullptr, int ldx_narrow = 0,
                       int cin_narrow = 0, const void* y_narrow = nullptr, float* dw_narrow = nullptr, float* db_narrow = nullptr);     // conv_wgrad_tr.hip
int tg_wgrad_bf16_multi_launch(const tg_conv_desc* const* ds, int n, const void* const* x, const int* ldxs, const void* const* y,
                               const int* ldys, float* const* dw, float* const* dbias, hipStream_t st);   // below
static bool tg_wgrad_bf16_setup(const tg_conv_desc* d, const void* x, int ldx, const void* y, int ldy, float* dw, float* dbias,
                                WgradBP& p, int& base_blocks);                                             // below

static bool tg_wgrad_row3_applies(const tg_conv_desc* d, int ldx, int ldy) {
  const bool enabled = true;
  if (!enabled || d->KW != 3 || d->stride != 1 || d->pad_l != 1 || d->Win != d->Wout || d->Hin != d->Hout) return false;
  if ((d->Wout & 1) || d->KH > 11 || d->pad_t < 0 || d->pad_t >= d->KH) return false;
  const int64_t M64 = (int64_t)d->N * d->Hout * d->Wout;
  return M64 * ldx < ((int64_t)1 << 29) && M64 * ldy < ((int64_t)1 << 29);                       // 32-bit byte offsets
}

// groups >= 1 layers, each with its own descriptor / channel strides (identical descriptors share a geometry entry); pointer
// arrays live on the host and are copied into the kernel arguments.  Returns 1 if launched, 0 if some layer does not fit.
static int tg_wgrad_row3_launch_multi(const tg_conv_desc* const* ds, int groups, const void* const* x, const int* ldxs,
                                      const void* const* y, const int* ldys, float* const* dw, float* const* dbias, hipStream_t st) {
  if (groups < 1 || groups > TG_WGRAD_MAX_GROUPS) return 0;
  WgradRP p;
  int ngeo = 0;
  double total_work = 0.0, flops = 0.0, bytes = 0.0;
  int base_blocks[TG_WGRAD_MAX_GEOS];
  double work_of[TG_WGRAD_MAX_GEOS];
  int members[TG_WGRAD_MAX_GEOS];
  for (int g = 0; g < groups; ++g) {
    const tg_conv_desc* d = ds[g];
    if (!tg_wgrad_row3_applies(d, ldxs[g], ldys[g])) return 0;
    const int64_t M64 = (int64_t)d->N * d->Hout * d->Wout;
    int e = -1;
    for (int k = 0; k < ngeo && e < 0; ++k) {
      const WgradGeo& q = p.geo[k];
      if (q.N == d->N && q.H == d->Hout && q.W == d->Wout && q.Cx == d->Cin && q.Cy == d->Cout && q.KH == d->KH && q.pt == d->pad_t &&
          q.ldx == ldxs[g] && q.ldy == ldys[g])
        e = k;
    }
    if (e < 0) {
      if (ngeo == TG_WGRAD_MAX_GEOS) return 0;
      e = ngeo++;
      WgradGeo& q = p.geo[e];
      q.N = d->N; q.H = d->Hout; q.W = d->Wout; q.Cx = d->Cin; q.Cy = d->Cout; q.KH = d->KH; q.pt = d->pad_t;
      q.M = (int)M64; q.ldx = ldxs[g]; q.ldy = ldys[g];
      q.xbytes = (unsigned)(M64 * ldxs[g] * 2);
      q.ybytes = (unsigned)(M64 * ldys[g] * 2);
      q.xtiles = (q.Cx + 63) / 64;
      q.ytiles = (q.Cy + 63) / 64;
      base_blocks[e] = d->KH * q.xtiles * q.ytiles;
      work_of[e] = (double)M64 * base_blocks[e];
      members[e] = 0;
    }
    members[e]++;
    p.gidx[g] = (unsigned char)e;
    total_work += work_of[e];
    flops += 2.0 * (double)M64 * 3.0 * d->KH * d->Cin * d->Cout;
    bytes += (double)M64 * (ldxs[g] + ldys[g]) * 2.0 + 12.0 * d->KH * d->Cin * d->Cout;
    p.xs[g] = (const u16*)x[g]; p.ys[g] = (const u16*)y[g]; p.dws[g] = dw[g]; p.dbs[g] = dbias ? dbias[g] : nullptr;
  }
  for (int g = groups; g < TG_WGRAD_MAX_GROUPS; ++g) {
    p.xs[g] = p.xs[0]; p.ys[g] = p.ys[0]; p.dws[g] = p.dws[0]; p.dbs[g] = p.dbs[0]; p.gidx[g] = 0;
  }
  // split-K degree per geometry.  One geometry (the old grouped form): the launch-wide model of tg_wgrad_ksplit.  Several:
  // every layer gets the share of ~768 workgroups its pixel x tile volume has in the launch (at least one per base block).
  for (int e = 0; e < ngeo; ++e) {
    WgradGeo& q = p.geo[e];
    int ksplit;
    if (ngeo == 1) {
      ksplit = tg_wgrad_ksplit(q.M, (int64_t)members[e] * q.KH * 3 * q.xtiles * q.ytiles * 4096, members[e] * base_blocks[e], 64);
    } else {
      ksplit = (int)(768.0 * work_of[e] / total_work / base_blocks[e] + 0.5);
      const int max_split = (q.M + 127) / 128;
      if (ksplit > max_split) ksplit = max_split;
      if (ksplit < 1 || tg_det()) ksplit = 1;
    }
    q.chunk = (((q.M + ksplit - 1) / ksplit) + 63) / 64 * 64;
    q.nchunk = (q.M + q.chunk - 1) / q.chunk;
  }
  int w = 0;
  for (int g = 0; g < groups; ++g) {
    p.wstart[g] = w;
    w += base_blocks[p.gidx[g]] * p.geo[p.gidx[g]].nchunk;
  }
  for (int g = groups; g <= TG_WGRAD_MAX_GROUPS; ++g) p.wstart[g] = w;
  p.groups = groups;
  // TG_CONV_COEXIST: 24 KB of unused dynamic LDS cap the residency at 2 workgroups per CU (instead of 4), so that a
  // workgroup of the latency-bound chain still finds registers on every CU while this launch runs beside it
  const unsigned pad_lds = (ds[0]->flags & TG_CONV_COEXIST) ? 24576u : 0u;
  TG_LAUNCH(ngeo == 1 ? "conv_wgrad_row3_bf16" : "conv_wgrad_row3_bf16_multi", flops, bytes, conv_wgrad_row3_bf16_kernel,
            dim3((unsigned)w), dim3(256), pad_lds, st, p);
  return 1;
}

// groups >= 1 layers of identical geometry
static int tg_wgrad_row3_launch(const tg_conv_desc* d, int groups, const void* const* x, int ldx, const void* const* y, int ldy,
                                float* const* dw, float* const* dbias, hipStream_t st) {
  if (groups < 1 || groups > TG_WGRAD_MAX_GROUPS) return 0;
  const tg_conv_desc* ds[TG_WGRAD_MAX_GROUPS];
  int lx[TG_WGRAD_MAX_GROUPS], ly[TG_WGRAD_MAX_GROUPS];
  for (int g = 0; g < groups; ++g) { ds[g] = d; lx[g] = ldx; ly[g] = ldy; }
  return tg_wgrad_row3_launch_multi(ds, groups, x, lx, y, ly, dw, dbias, st);
}

static int tg_wgrad_row3_try(const tg_conv_desc* d, const void* x, int ldx, const void* y, int ldy, float* dw, float* dbias,
                             hipStream_t st) {
  return tg_wgrad_row3_launch(d, 1, &x, ldx, &y, ldy, &dw, dbias ? &dbias : nullptr, st);
}

extern "C" int tg_conv_wgrad_grouped(const tg_conv_desc* d, int groups, const void* const* x, int x_dtype, int ldx,
                                     const void* const* y, int y_dtype, int ldy, float* const* dw, float* const* dbias,
                                     void* stream) {
  TG_CHECK_ARG(d && x && y && dw && groups >= 1, "null pointer / no groups");
  for (int g = 0; g < groups; ++g) TG_CHECK_ARG(x[g] && y[g] && dw[g], "null group pointer");
  const int lx = ldx > 0 ? ldx : d->Cin, ly = ldy > 0 ? ldy : d->Cout;
  bool fast = d->mode == 0 && x_dtype == TG_BF16 && y_dtype == TG_BF16 && lx % 8 == 0 && ly % 8 == 0 && lx >= d->Cin &&
              ly >= d->Cout;
  for (int g = 0; g < groups && fast; ++g) fast = ((((uintptr_t)x[g] | (uintptr_t)y[g])) & 15) == 0;
  if (fast && tg_wgrad_tr_launch(d, groups, x, lx, y, ly, dw, dbias, static_cast<hipStream_t>(stream))) TG_CHECK_LAUNCH();
  if (fast && tg_wgrad_row3_launch(d, groups, x, lx, y, ly, dw, dbias, static_cast<hipStream_t>(stream)))
    TG_CHECK_LAUNCH();
  for (int g = 0; g < groups; ++g) {                        // any other geometry / dtype: one ordinary launch per layer
    const int rc = tg_conv_wgrad(d, x[g], x_dtype, ldx, y[g], y_dtype, ldy, dw[g], dbias ? dbias[g] : nullptr, stream);
    if (rc != TG_OK) return rc;
  }
  return TG_OK;
}

// tg_conv_wgrad_grouped plus ONE more layer of the same spatial geometry and output width but FEWER input channels (the generator's
// input conv beside its residual trunk): one launch when the transpose-read kernel takes the group, otherwise the grouped call and
// an ordinary tg_conv_wgrad for the extra layer.
extern "C" int tg_conv_wgrad_grouped_plus(const tg_conv_desc* d, int groups, const void* const* x, int x_dtype, int ldx,
                                          const void* const* y, int y_dtype, int ldy, float* const* dw, float* const* dbias,
                                          const void* x_extra, int ldx_extra, int cin_extra, const void* y_extra, float* dw_extra,
                                          float* dbias_extra, void* stream) {
  TG_CHECK_ARG(d && x && y && dw && groups >= 1 && x_extra && y_extra && dw_extra, "null pointer / no groups");
  TG_CHECK_ARG(cin_extra >= 1 && cin_extra <= d->Cin && ldx_extra >= cin_extra, "the extra layer has at most the group's input channels");
  for (int g = 0; g < groups; ++g) TG_CHECK_ARG(x[g] && y[g] && dw[g], "null group pointer");
  const int lx = ldx > 0 ? ldx : d->Cin, ly = ldy > 0 ? ldy : d->Cout;
  bool fast = d->mode == 0 && x_dtype == TG_BF16 && y_dtype == TG_BF16 && lx % 8 == 0 && ly % 8 == 0 && lx >= d->Cin && ly >= d->Cout;
  for (int g = 0; g < groups && fast; ++g) fast = ((((uintptr_t)x[g] | (uintptr_t)y[g])) & 15) == 0;
  if (fast && tg_wgrad_tr_launch(d, groups, x, lx, y, ly, dw, dbias, static_cast<hipStream_t>(stream), x_extra, ldx_extra, cin_extra,
                                 y_extra, dw_extra, dbias_extra))
    TG_CHECK_LAUNCH();
  const int rc = tg_conv_wgrad_grouped(d, groups, x, x_dtype, ldx, y, y_dtype, ldy, dw, dbias, stream);
  if (rc != TG_OK) return rc;
  tg_conv_desc de = *d;
  de.Cin = cin_extra;
  return tg_conv_wgrad(&de, x_extra, x_dtype, ldx_extra, y_extra, y_dtype, ldy, dw_extra, dbias_extra, stream);
}

// Weight gradients of `groups` layers of DIFFERENT geometry in one call: descs[g] / ldx[g] / ldy[g] per layer.  bf16 3x3 stride-1
// SAME layers (FNet's 14 convs, reference lib/frvsr.py:4-41 under tf.gradients) go out as ONE launch; anything else falls
// back to one tg_conv_wgrad per layer.
extern "C" int tg_conv_wgrad_multi(const tg_conv_desc* descs, int groups, const void* const* x, int x_dtype, const int* ldx,
                                   const void* const* y, int y_dtype, const int* ldy, float* const* dw, float* const* dbias,
                                   void* stream) {
  TG_CHECK_ARG(descs && x && y && dw && ldx && ldy && groups >= 1, "null pointer / no groups");
  // layers the row kernel takes go out together; every other one (dtype, stride, odd width, unpadded channel strides ...) alone
  const tg_conv_desc* ds[TG_WGRAD_MAX_GROUPS];
  const void *xe[TG_WGRAD_MAX_GROUPS], *ye[TG_WGRAD_MAX_GROUPS];
  float *we[TG_WGRAD_MAX_GROUPS], *be[TG_WGRAD_MAX_GROUPS];
  int lx[TG_WGRAD_MAX_GROUPS], ly[TG_WGRAD_MAX_GROUPS], ne = 0;
  bool taken[TG_WGRAD_MAX_GROUPS] = {false};
  for (int g = 0; g < groups; ++g) {
    TG_CHECK_ARG(x[g] && y[g] && dw[g], "null group pointer");
    if (x_dtype != TG_BF16 || y_dtype != TG_BF16 || groups > TG_WGRAD_MAX_GROUPS) continue;
    const tg_conv_desc* d = descs + g;
    const int a = ldx[g] > 0 ? ldx[g] : d->Cin, b = ldy[g] > 0 ? ldy[g] : d->Cout;
    if (d->mode == 0 && a % 8 == 0 && b % 8 == 0 && a >= d->Cin && b >= d->Cout && ((((uintptr_t)x[g] | (uintptr_t)y[g])) & 15) == 0 &&
        tg_wgrad_row3_applies(d, a, b)) {
      ds[ne] = d; xe[ne] = x[g]; ye[ne] = y[g]; we[ne] = dw[g]; be[ne] = dbias ? dbias[g] : nullptr; lx[ne] = a; ly[ne] = b;
      taken[g] = true;
      ++ne;
    }
  }
  if (ne >= 2) {
    bool any_bias = false;
    for (int k = 0; k < ne; ++k) any_bias = any_bias || be[k] != nullptr;
    if (!tg_wgrad_row3_launch_multi(ds, ne, xe, lx, ye, ly, we, any_bias ? be : nullptr, static_cast<hipStream_t>(stream)))
      for (int g = 0; g < groups && g < TG_WGRAD_MAX_GROUPS; ++g) taken[g] = false;
    else if (hipGetLastError() != hipSuccess) { tg_set_error("%s: launch failed", __func__); return TG_ELAUNCH; }
  } else {
    for (int g = 0; g < groups && g < TG_WGRAD_MAX_GROUPS; ++g) taken[g] = false;     // (groups > the table: `taken` has 40 entries)
  }
  // the layers left over that the per-tap bf16 kernel takes (strided / non-3-wide kernels: the discriminator's 4x4 stride-2 convs)
  // go out as ONE multi-layer launch of that kernel, largest layer first
  if (x_dtype == TG_BF16 && y_dtype == TG_BF16 && groups <= TG_WGRAD_MAX_GROUPS) {
    int idx[TG_WGRAD_BM_MAX], nb = 0;
    for (int g = 0; g < groups && nb < TG_WGRAD_BM_MAX; ++g) {
      if (taken[g]) continue;
      const tg_conv_desc* d = descs + g;
      const int a = ldx[g] > 0 ? ldx[g] : d->Cin, b = ldy[g] > 0 ? ldy[g] : d->Cout;
      WgradBP tmp;
      int bb;
      // (3x3 s1 layers the row kernel takes alone stay with it: tg_conv_wgrad below)
      if (d->mode == 0 && a >= d->Cin && b >= d->Cout && !tg_wgrad_row3_applies(d, a, b) &&
          tg_wgrad_bf16_setup(d, x[g], a, y[g], b, dw[g], nullptr, tmp, bb))
        idx[nb++] = g;
    }
    if (nb >= 2) {
      for (int i = 1; i < nb; ++i)                      // insertion sort by work (pixels x taps x channel tiles), descending
        for (int j = i; j > 0; --j) {
          const tg_conv_desc *da = descs + idx[j - 1], *db = descs + idx[j];
          const double wa = (double)da->N * da->Hout * da->Wout * da->KH * da->KW * ((da->Cin + 63) / 64) * ((da->Cout + 63) / 64);
          const double wb = (double)db->N * db->Hout * db->Wout * db->KH * db->KW * ((db->Cin + 63) / 64) * ((db->Cout + 63) / 64);
          if (wb > wa) { const int t = idx[j]; idx[j] = idx[j - 1]; idx[j - 1] = t; } else break;
        }
      const tg_conv_desc* bd[TG_WGRAD_BM_MAX];
      const void *bx[TG_WGRAD_BM_MAX], *by[TG_WGRAD_BM_MAX];
      float *bw[TG_WGRAD_BM_MAX], *bb2[TG_WGRAD_BM_MAX];
      int blx[TG_WGRAD_BM_MAX], bly[TG_WGRAD_BM_MAX];
      bool any_bias = false;
      for (int i = 0; i < nb; ++i) {
        const int g = idx[i];
        bd[i] = descs + g; bx[i] = x[g]; by[i] = y[g]; bw[i] = dw[g]; bb2[i] = dbias ? dbias[g] : nullptr;
        blx[i] = ldx[g] > 0 ? ldx[g] : descs[g].Cin; bly[i] = ldy[g] > 0 ? ldy[g] : descs[g].Cout;
        any_bias = any_bias || bb2[i] != nullptr;
      }
      if (tg_wgrad_bf16_multi_launch(bd, nb, bx, blx, by, bly, bw, any_bias ? bb2 : nullptr, static_cast<hipStream_t>(stream))) {
        if (hipGetLastError() != hipSuccess) { tg_set_error("%s: launch failed", __func__); return TG_ELAUNCH; }
        for (int i = 0; i < nb; ++i) taken[idx[i]] = true;
      }
    }
  }
  for (int g = 0; g < groups; ++g) {
    if (g < TG_WGRAD_MAX_GROUPS && taken[g]) continue;
    const int rc = tg_conv_wgrad(descs + g, x[g], x_dtype, ldx[g], y[g], y_dtype, ldy[g], dw[g], dbias ? dbias[g] : nullptr, stream);
    if (rc != TG_OK) return rc;
  }
  return TG_OK;
}

// Parameters of the per-tap bf16 kernel for one layer; false if the layer does not fit it (then: the generic kernel).
static bool tg_wgrad_bf16_setup(const tg_conv_desc* d, const void* x, int ldx, const void* y, int ldy, float* dw, float* dbias,
                                WgradBP& p, int& base_blocks) {
  if (ldx % 8 || ldy % 8 || (((uintptr_t)x | (uintptr_t)y) & 15)) return false;
  p.x = (const u16*)x; p.y = (const u16*)y; p.dw = dw; p.dbias = dbias;
  p.N = d->N; p.Hx = d->Hin; p.Wx = d->Win; p.Cx = d->Cin; p.Hy = d->Hout; p.Wy = d->Wout; p.Cy = d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.s = d->stride; p.pt = d->pad_t; p.pl = d->pad_l;
  p.M = d->N * d->Hout * d->Wout; p.ldx = ldx; p.ldy = ldy;
  const int64_t M64 = (int64_t)d->N * d->Hout * d->Wout;
  if (M64 >= ((int64_t)1 << 30) || (int64_t)d->N * d->Hin * d->Win * ldx >= ((int64_t)1 << 30) ||
      M64 * ldy >= ((int64_t)1 << 30) || (d->Wout & 1))
    return false;                       // 32-bit byte offsets out of range, or odd width (pixel pairs): generic kernel
  p.xbytes = (unsigned)((int64_t)d->N * d->Hin * d->Win * ldx * 2);
  p.ybytes = (unsigned)(M64 * ldy * 2);
  p.xtiles = (p.Cx + 63) / 64;
  p.ytiles = (p.Cy + 63) / 64;
  base_blocks = d->KH * d->KW * p.xtiles * p.ytiles;
  return true;
}

// returns 1 if launched
int tg_wgrad_bf16_try(const tg_conv_desc* d, const void* x, int x_dtype, int ldx, const void* y, int y_dtype, int ldy,
                      float* dw, float* dbias, hipStream_t st) {
  if (x_dtype != TG_BF16 || y_dtype != TG_BF16) return 0;
  if (ldx % 8 || ldy % 8 || (((uintptr_t)x | (uintptr_t)y) & 15)) return 0;
  if (d->mode == 0 && ldx >= d->Cin && ldy >= d->Cout && tg_wgrad_tr_launch(d, 1, &x, ldx, &y, ldy, &dw, dbias ? &dbias : nullptr, st))
    return 1;                           // 64-channel 3x3 layers on images whose width is a multiple of 32 (conv_wgrad_tr.hip)
  if (tg_wgrad_row3_try(d, x, ldx, y, ldy, dw, dbias, st)) return 1;
  WgradBP p;
  int base_blocks = 0;
  if (!tg_wgrad_bf16_setup(d, x, ldx, y, ldy, dw, dbias, p, base_blocks)) return 0;
  const int64_t M64 = (int64_t)d->N * d->Hout * d->Wout;
  const int pf = 2;                                              // 2: best or within noise at every swept shape (round 2)
  const int quantum = 64 * pf;
  int ksplit = tg_wgrad_ksplit(p.M, (int64_t)base_blocks * 4096, base_blocks, quantum);
  p.chunk = (((p.M + ksplit - 1) / ksplit) + quantum - 1) / quantum * quantum;
  ksplit = (p.M + p.chunk - 1) / p.chunk;
  const dim3 grid((unsigned)(base_blocks * ksplit));   // 1-D: the kernel maps work XCD-aware
  const unsigned lds = 2u * 64u * (128u * pf + 8u) + ((d->flags & TG_CONV_COEXIST) ? 24576u : 0u);  // see row3
  const double wfl = 2.0 * (double)M64 * d->KH * d->KW * (double)d->Cin * d->Cout;
  const double wby = (double)d->N * d->Hin * d->Win * ldx * 2.0 + (double)M64 * ldy * 2.0 + 4.0 * d->KH * d->KW * d->Cin * d->Cout;
  TG_LAUNCH("conv_wgrad_bf16<2>", wfl, wby, conv_wgrad_bf16_kernel<2>, grid, dim3(256), lds, st, p);
  return 1;
}

// `n` layers (2 <= n <= TG_WGRAD_BM_MAX) the per-tap kernel takes, as ONE launch.  Split-K per layer: the layer with the longest
// per-workgroup chain alone (steps / its own optimal split) sets the chain length T of the launch; every other layer is split so
// that its workgroups run about T steps too, never finer than it would be split alone (fewer, equally long workgroups: the small
// layers' atomics shrink with their split and their workgroups fill the large layer's tail).  Returns 1 if launched.
int tg_wgrad_bf16_multi_launch(const tg_conv_desc* const* ds, int n, const void* const* x, const int* ldxs, const void* const* y,
                               const int* ldys, float* const* dw, float* const* dbias, hipStream_t st) {
  if (n < 2 || n > TG_WGRAD_BM_MAX) return 0;
  WgradBM P;
  int base[TG_WGRAD_BM_MAX], alone[TG_WGRAD_BM_MAX];
  constexpr int quantum = 128;
  double T = 1.0, flops = 0.0, bytes = 0.0;
  for (int g = 0; g < n; ++g) {
    const tg_conv_desc* d = ds[g];
    if (d->mode != 0 || ldxs[g] < d->Cin || ldys[g] < d->Cout) return 0;
    if (!tg_wgrad_bf16_setup(d, x[g], ldxs[g], y[g], ldys[g], dw[g], dbias ? dbias[g] : nullptr, P.g[g], base[g])) return 0;
    alone[g] = tg_wgrad_ksplit(P.g[g].M, (int64_t)base[g] * 4096, base[g], quantum);
    const double t = (double)P.g[g].M / 64.0 / alone[g];
    if (t > T) T = t;
    const double M = (double)P.g[g].M;
    flops += 2.0 * M * d->KH * d->KW * (double)d->Cin * d->Cout;
    bytes += (double)d->N * d->Hin * d->Win * ldxs[g] * 2.0 + M * ldys[g] * 2.0 + 4.0 * d->KH * d->KW * d->Cin * d->Cout;
  }
  int w = 0;
  for (int g = 0; g < n; ++g) {
    WgradBP& p = P.g[g];
    int ksplit = (int)((double)p.M / 64.0 / T + 0.5);
    if (ksplit > alone[g]) ksplit = alone[g];
    if (ksplit < 1 || tg_det()) ksplit = 1;
    p.chunk = (((p.M + ksplit - 1) / ksplit) + quantum - 1) / quantum * quantum;
    ksplit = (p.M + p.chunk - 1) / p.chunk;
    P.wstart[g] = w;
    w += base[g] * ksplit;
  }
  for (int g = n; g < TG_WGRAD_BM_MAX; ++g) P.g[g] = P.g[0];
  for (int g = n; g <= TG_WGRAD_BM_MAX; ++g) P.wstart[g] = w;
  P.groups = n;
  const unsigned lds = 2u * 64u * (128u * 2 + 8u) + ((ds[0]->flags & TG_CONV_COEXIST) ? 24576u : 0u);
  TG_LAUNCH("conv_wgrad_bf16_multi", flops, bytes, conv_wgrad_bf16_multi_kernel, dim3((unsigned)w), dim3(256), lds, st, P);
  return 1;
}
